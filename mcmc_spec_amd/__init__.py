"""MI355X-native log-likelihood / log-posterior path of mft6.py."""
import os as _os

# Kernel arguments in device memory: with host-resident kernargs every workgroup's first scalar loads cross
# PCIe and the 20 us hot kernel takes 26 us (measured, HIP_FORCE_DEV_KERNARG=0 vs 1).  It is the default on this
# platform; pinned here, before the HIP runtime initialises, so that an inherited environment cannot undo it.
_os.environ.setdefault('HIP_FORCE_DEV_KERNARG', '1')
