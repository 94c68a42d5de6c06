#!/usr/bin/env python3
"""Headline benchmark: walker log-posterior evaluations per second (BASELINE.json `metric`).

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
    python bench.py --config 5            # 8 KOI targets, one independent problem per GPU (replicas, no collective)

One "step" = one fused launch of the per-walker log-posterior (prior gate + two-component spectrum synthesis +
reddening + resample + median/continuum normalisation + chi^2 + contrast terms) over this rank's walkers, inputs
resident in HBM, followed -- when N > 1 -- by one RCCL all-gather of the log-probabilities (walkers are
independent: shard, no other data-path collective).  Workload at N = 1 is BASELINE.json configs[1]: binary,
4096-pixel spectrum, 256 walkers; weak scaling keeps 256 walkers per GPU (N = 8 is configs[2], 2048 walkers).

Rank 0 prints ONE JSON line (contract in the task statement) including `roofline`, `cpu_baseline` and, at N = 1,
`extra`: the batch-size sweep, BASELINE config 4's per-GPU share and the CPU baseline at the reference's own pool
width -- every number DESIGN.md quotes comes out of this line or of a file under profiles/.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

# MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec; L2 ~34.5 TB/s aggregate (66-73 GB/s per CU measured for row gathers
# served by the XCD's L2); FP32 vector peak 157.3 TFLOP/s -- FP64 vector issues at half that rate (16 lanes per
# clock per SIMD: 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz = 78.6 TFLOP/s).
HBM_PEAK_GBPS = 8000.0
L2_PEAK_GBPS = 34500.0
FP64_VECTOR_PEAK_TFLOPS = 78.6


def build_workload(eng, npix, with_phot, resolution=1700, seed=2, keep_host_grid=False, grid=None, store='f64', broaden='staging'):
    """Stage the synthetic 26x4x135,000 grid, broaden the data window on the device (A3), synthesise
    a data spectrum at theta* with the GPU's own make_composite and stage the problem."""
    from mcmc_spec_amd import bands, staging, synth
    wl = np.arange(3000, 30000, 0.2)  # mft6.py:343 with specmin/specmax of param_koi2298.txt:15-16
    teffs = np.arange(3000, 5600, 100)
    loggs = np.array([4.0, 4.5, 5.0, 5.5])
    flux = grid if grid is not None else synth.make_grid(teffs, loggs, wl)
    eng.stage_grid(wl, teffs, loggs, flux)
    wl_um = synth.data_wavelengths_um(npix)
    r = [float(wl_um.min()), float(wl_um.max())]
    win = [np.floor(r[0] * 1e4), np.ceil(r[1] * 1e4)]  # "spmin/spmax" of the run, Angstrom
    eng.broaden_grid_window(win, resolution, broaden)   # ('in_path': the raw window stays on the device beside the broadened grid)
    matrix = synth.make_isochrone_matrix()
    ctm = synth.synthetic_contrast_filters()
    if with_phot:
        ptm = synth.synthetic_phot_filters()
        pfilt, pmag, perr = ['sdss,r', 'sdss,i', 'sdss,z', 'j', 'h', 'k'], synth.EXAMPLE_PMAG, synth.EXAMPLE_PERR
    else:
        ptm, pfilt, pmag, perr = [[], [], [], []], [], [], []
    fr = [synth.EXAMPLE_CMAG, synth.EXAMPLE_CERR, ['lp600', 'Kp'], pmag, perr, pfilt]
    tmi = min(min(w) for w in ctm[0] + ptm[0])
    tma = max(max(w) for w in ctm[0] + ptm[0])
    tabs, (vw, vf) = synth.synthetic_band_tables(), synth.synthetic_vega()
    bl = bands.make_bands(tabs, vw, vf)
    av_table = synth.make_av_table()
    prior = [*np.zeros(10), 2.0732e-3, 0.0277e-3]  # mft6.py:3689
    kw = dict(nspec=2, bands=bl, av_table=av_table, tmin=float(teffs[0]), tmax=float(teffs[-1]), prior=prior)
    # pass 1: placeholder data so that make_composite has its window/tables; pass 2: real data
    ones = np.ones(npix)
    eng.stage_problem([wl_um, ones], ones, fr, r, ctm, ptm, tmi, tma, matrix, **kw)
    p = synth.TRUTH_THETA
    lg = staging.isochrone_logg(p[:2], matrix)
    w1, c1, _, _, _ = eng.make_composite(p[:2], lg, p[3:5], p[5])
    c1 = c1 * 10.0 ** (-0.4 * p[2] * eng.ctx.ccm89_k(w1, 3.1))  # extinct(), mft6.py:62-63
    f = np.interp(wl_um * 1e4, w1, c1)
    rng = np.random.default_rng(seed)
    d = f + rng.normal(0, 0.01 * f)  # mft6.py:3640
    med = np.median(d)
    data, err = [wl_um, d / med], 0.01 * f / med  # mft6.py:3506-3507
    eng.stage_problem(data, err, fr, r, ctm, ptm, tmi, tma, matrix, store=store, **kw)
    nwin = int(np.sum((wl >= win[0] - 1) & (wl <= win[1] + 1)))
    out = dict(data=data, err=err, fr=fr, r=r, ctm=ctm, ptm=ptm, tmi=tmi, tma=tma, matrix=matrix, tabs=tabs,
               vega=(vw, vf), prior=prior, tmin=float(teffs[0]), tmax=float(teffs[-1]), nwin=nwin, win=win,
               resolution=resolution, teffs=teffs, loggs=loggs, wl=wl)
    if keep_host_grid:
        out['flux'] = flux
    return out


def build_koi_problem(eng, target_index, grid=None):
    """BASELINE config 5: one of the eight KOI spectra (prepared exactly like mft6.py:3492-3507; the prepared
    vectors travel as the committed fixture tests/golden/golden_koi.npz -- /root/reference does not exist on the
    GPU box) on the full-size synthetic grid, contrast terms through the real lp600 / Kp tables of the fixture."""
    from mcmc_spec_amd import bands, synth
    g = np.load(os.path.join(ROOT, 'tests', 'golden', 'golden_koi.npz'))
    gr = np.load(os.path.join(ROOT, 'tests', 'golden', 'golden_reference.npz'))
    tag = str(g['targets'][target_index % len(g['targets'])])
    wl = np.arange(3000, 30000, 0.2)
    teffs = np.arange(3000, 5600, 100)
    loggs = np.array([4.0, 4.5, 5.0, 5.5])
    flux = grid if grid is not None else synth.make_grid(teffs, loggs, wl)
    eng.stage_grid(wl, teffs, loggs, flux)
    data, err = [g[tag + '_wl'], g[tag + '_flux']], g[tag + '_err']
    r = [float(min(data[0])), float(max(data[0]))]
    eng.broaden_grid_window([np.floor(0.55 * 1e4), np.ceil(0.90 * 1e4)], 1700)  # spmin / spmax of the crop
    ctm = [[list(gr['ctmA_w0']), list(gr['ctmA_w1'])], [list(gr['ctmA_t0']), list(gr['ctmA_t1'])], [0, 0],
           [np.mean(gr['ctmA_w0']), np.mean(gr['ctmA_w1'])]]
    ptm = [[], [], [], []]
    fr = [synth.EXAMPLE_CMAG, synth.EXAMPLE_CERR, ['lp600', 'Kp'], [], [], []]
    tmi = min(min(w) for w in ctm[0])
    tma = max(max(w) for w in ctm[0])
    tabs, (vw, vf) = synth.synthetic_band_tables(), synth.synthetic_vega()
    prior = [*np.zeros(10), 2.0732e-3, 0.0277e-3]
    eng.stage_problem(data, err, fr, r, ctm, ptm, tmi, tma, synth.make_isochrone_matrix(), nspec=2,
                      bands=bands.make_bands(tabs, vw, vf), av_table=synth.make_av_table(), tmin=float(teffs[0]),
                      tmax=float(teffs[-1]), prior=prior, rad_prior=True)
    return dict(tag=tag, npix=len(err), tmin=float(teffs[0]), tmax=float(teffs[-1]), flux=flux)


# ------------------------------------------------------------------------------------------------
# CPU baseline: the oracle driven like emcee drives a pool.  Only this leg of
# bench.py touches oracle/.
# ------------------------------------------------------------------------------------------------
_CPU = {}


def _cpu_one(theta):
    from oracle import mft6_oracle as orc
    W = _CPU['w']
    return orc.logposterior(list(theta), W['fr'], 2, W['data'], W['err'], W['r'], _CPU['specs'], W['ctm'], W['ptm'],
                            W['tmi'], W['tma'], W['tmin'], W['tmax'], W['matrix'], _CPU['av_prior'], prior=W['prior'],
                            bandlib=_CPU['bandlib'])


def cpu_baseline(W, theta, gpu_logp, budget_s=20.0, procs=(0,)):
    """Oracle `logposterior` under multiprocessing.Pool(P) (fork; tables inherited) for each P in `procs`
    (0 = this job's CPU share, at most 16).  Returns one record per P; the first is `cpu_baseline`."""
    import multiprocessing as mp
    import warnings
    from mcmc_spec_amd import synth
    from oracle import mft6_oracle as orc
    warnings.filterwarnings('ignore')
    specs = synth.grid_to_specs(W['teffs'], W['loggs'], W['wl'], W['flux'])
    specs = orc.broaden_specs_window(specs, W['win'], W['resolution'])  # CPU restatement of the A3 staging
    edges, mu, sig = synth.make_av_table()

    def av_prior(d):
        b = int(np.clip(np.searchsorted(edges, d, side='right') - 1, 0, len(mu) - 1))
        return mu[b], sig[b]

    _CPU.update(w=W, specs=specs, av_prior=av_prior, bandlib=orc.make_band_library(W['tabs'], *W['vega']))
    t0 = time.time()
    _cpu_one(theta[0])
    t_one = time.time() - t0
    out = []
    for pw in procs:
        # the GPU box gives one GPU's CPU share (16 cores) to this job; never oversubscribe it
        cores = pw or min(len(os.sched_getaffinity(0)), 16)
        n = int(max(cores, min(len(theta), budget_s * cores / max(t_one, 1e-3))))
        n = min(n, len(theta))
        ctx = mp.get_context('fork')  # tables inherited, not pickled per call
        with ctx.Pool(processes=cores) as pool:
            pool.map(_cpu_one, theta[:cores])  # warm-up
            t0 = time.time()
            res = pool.map(_cpu_one, theta[:n])
            dt = time.time() - t0
        res = np.array(res)
        g = gpu_logp[:n]
        fin = np.isfinite(res)
        rel = float(np.max(np.abs(res[fin] - g[fin]) / np.abs(res[fin]))) if fin.any() else 0.0
        same_inf = bool(np.array_equal(np.isinf(res), np.isinf(g)))
        out.append(dict(value=n / dt, unit='evals/s', cores=cores, kind='port',
                        sample='{} walkers of the same ensemble, oracle logposterior under multiprocessing.Pool({}) '
                               '(fork); single eval {:.1f} ms'.format(n, cores, t_one * 1e3),
                        max_rel_err_gpu_vs_oracle=rel, inf_pattern_equal=same_inf))
    return out


def device_time_us(eng, theta_dev, lp, st, stream, n, iters, warm=3, reps=1):
    """Mean device time of one launch over `n` walkers (HIP events on the launch stream around `iters` launches; the
    median of `reps` such blocks behind `warm` untimed launches)."""
    import torch
    from mcmc_spec_amd import _lib
    def go():
        eng.ctx.logprob_batch_dev(theta_dev.data_ptr(), n, 6, lp.data_ptr(), st.data_ptr(), stream.cuda_stream,
                                  _lib.MODE_LOGPOST, 0)
    for _ in range(warm):
        go()
    got = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(iters):
            go()
        e1.record(stream)
        torch.cuda.synchronize()
        got.append(e0.elapsed_time(e1) / iters * 1e3)
    return sorted(got)[len(got) // 2]


def load_profile(name):
    p = os.path.join(ROOT, 'profiles', name)
    return json.load(open(p)) if os.path.exists(p) else None


def self_launch(n_gpus, argv, run=None):
    """`python bench.py --gpus N ...` outside any launcher: run the same command under
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P` (one rank
    per GPU over RCCL -- the form the task statement gives for N > 1; the reference's own fan-out is the commented
    `threads=nwalkers` pool of /root/reference/mft6.py:1490-1492) as a CHILD process -- this process has not imported
    torch nor touched a GPU, and never replaces itself -- and pass rank 0's ONE JSON line through on stdout.
    Returns the child's return code (non-zero too when it printed no result line)."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s_:
        s_.bind(('127.0.0.1', 0))
        port = s_.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n_gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    print('[bench] --gpus {} without WORLD_SIZE: launching {}'.format(n_gpus, ' '.join(cmd)), file=sys.stderr, flush=True)
    res = (run or subprocess.run)(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = [ln for ln in (res.stdout or '').splitlines() if ln.lstrip().startswith('{')]
    for ln in (res.stdout or '').splitlines():
        if not ln.lstrip().startswith('{') and ln.strip():
            print(ln, file=sys.stderr)     # anything else a rank or the launcher wrote to stdout
    if lines:
        sys.stdout.write(lines[-1].strip() + '\n')
        sys.stdout.flush()
    if res.returncode != 0:
        return res.returncode
    return 0 if lines else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--config', type=int, default=2, choices=[2, 3, 4, 5],
                    help='BASELINE config: 2/3 = 4096 px + contrast terms (default), 4 = 16384 px + 6-band photometry '
                         '(128 walkers per GPU), 5 = the eight KOI targets, one independent problem per GPU (512 walkers)')
    ap.add_argument('--walkers', type=int, default=0, help='walkers per GPU per launch (0 = the config\'s own)')
    ap.add_argument('--npix', type=int, default=0)
    ap.add_argument('--phot', action='store_true', help='add the 6-band photometry term')
    ap.add_argument('--block', type=int, default=0, help='threads per workgroup (0 = auto)')
    ap.add_argument('--store', default='f64', choices=['f64', 'f32'],
                    help='storage precision of the staged grid table R (f32: a separately labelled precision, never the headline)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-extras', action='store_true', help='skip the sweep / config-4 / P=15 extras of the N = 1 run')
    ap.add_argument('--no-overlap', action='store_true',
                    help='N > 1: wait for each all-gather before the next launch (a single dependent chain)')
    ap.add_argument('--cpu-budget', type=float, default=12.0)
    ap.add_argument('--copy-gib', type=float, default=1.0)
    args = ap.parse_args()
    if args.config == 4:
        args.npix, args.phot, args.walkers = args.npix or 16384, True, args.walkers or 128
    if args.config == 5:
        args.walkers = args.walkers or 512
    args.npix = args.npix or 4096
    args.walkers = args.walkers or 256
    # `python bench.py --gpus N` with N > 1 and no launcher around it: start the N ranks ourselves (a CHILD process,
    # before this one imports torch or touches a GPU), hand rank 0's one JSON line through and leave with its code
    # (MSX_BENCH_SELF_LAUNCH=1: the same for N = 1 -- the one-GPU rehearsal of this path, tests/test_gpu_bench_contract.py)
    if (args.gpus > 1 or os.environ.get('MSX_BENCH_SELF_LAUNCH') == '1') and 'WORLD_SIZE' not in os.environ:
        raise SystemExit(self_launch(args.gpus, sys.argv[1:]))

    # stdout carries exactly ONE line, the result: libraries that print banners there (RCCL does at start-up) are
    # sent to stderr for the duration of the run
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    from mcmc_spec_amd import _lib, synth
    from mcmc_spec_amd.engine import Engine

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit('--gpus {} but WORLD_SIZE={}: launch with torch.distributed.run'.format(args.gpus, world))
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    force_gather = os.environ.get('MSX_BENCH_FORCE_GATHER') == '1'  # measure collective overhead on one GPU
    replicas = args.config == 5                                      # independent problems: no collective at all
    if world > 1 or force_gather:
        if force_gather and 'RANK' not in os.environ:   # the one-GPU rehearsal outside any launcher: a one-rank group
            import socket
            with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s_:
                s_.bind(('127.0.0.1', 0))
                free_port = s_.getsockname()[1]
            os.environ.update(RANK='0', WORLD_SIZE='1', MASTER_ADDR='127.0.0.1', MASTER_PORT=str(free_port))
        dist.init_process_group('nccl', device_id=dev)

    n = args.walkers
    ndim = 6
    want_cpu = (rank == 0 and world == 1 and not args.no_cpu_baseline and not replicas)
    if replicas:
        # one problem per rank; on ONE GPU the rehearsal keeps all eight staged problems resident and cycles them
        nprob = 8 if world == 1 else 1
        engines, infos, grid = [], [], None
        for k in range(nprob):
            e = Engine(local)
            info = build_koi_problem(e, rank if world > 1 else k, grid=grid)
            grid = info.pop('flux')
            engines.append(e)
            infos.append(info)
        eng = engines[0]
        W = dict(tmin=infos[0]['tmin'], tmax=infos[0]['tmax'], nwin=None)
        args.npix = infos[0]['npix']
    else:
        eng = Engine(local)
        engines = [eng]
        W = build_workload(eng, args.npix, args.phot, keep_host_grid=True, store=args.store)
    # distinct coordinates per rank and a few distinct batches so no launch repeats the previous one
    nbatch = 4
    thetas = [torch.from_numpy(synth.draw_walkers(n, seed=3 + 1000 * rank + b, tmin=W['tmin'], tmax=W['tmax'])).to(dev)
              for b in range(nbatch)]
    # two output buffers: with N > 1 the all-gather of step i overlaps the kernel of step i+1 (the
    # collective runs on RCCL's stream; a buffer is only reused after its all-gather has completed)
    logp = [torch.empty(n, dtype=torch.float64, device=dev) for _ in range(2)]
    status = [torch.zeros(n, dtype=torch.int32, device=dev) for _ in range(2)]
    use_gather = (world > 1 or force_gather) and not replicas
    gathered = [torch.empty(n * world, dtype=torch.float64, device=dev) for _ in range(2)] if use_gather else None
    works = [None, None]
    stream = torch.cuda.current_stream(dev)
    sptr = stream.cuda_stream

    # the launch is a plain C call with pre-built arguments (no per-step Python marshalling)
    import ctypes as C
    fn = eng.ctx.lib.msx_logprob_batch_dev
    # With a collective in flight RCCL's kernel holds a CU or two.  The N = 1 variant (512 threads + 136 KB of LDS: one
    # workgroup per CU, all 256 CUs needed at once) would run a second round for the displaced walkers, so N > 1
    # launches the <= 128-VGPR variant, two of whose workgroups fit a CU (MSX_BLOCK_512_SHARED).  Same bits.
    block = args.block if args.block else (_lib.BLOCK_512_SHARED if use_gather and args.npix < 8192 else 0)

    def calls_for(sp):
        # step i: problem i mod nprob (config 5 rehearsal), theta batch i mod nbatch, output buffer i mod 2
        return [[[(e.ctx.h, _lib.MODE_LOGPOST, C.c_void_p(t.data_ptr()), n, ndim, C.c_void_p(logp[b].data_ptr()),
                   C.c_void_p(status[b].data_ptr()), C.c_void_p(sp), block) for b in range(2)] for t in thetas]
                for e in engines]

    calls = calls_for(sptr)
    nprob = len(engines)

    def launch(i, table=None):
        if fn(*(table or calls)[i % nprob][i % nbatch][i & 1]) != 0:
            raise RuntimeError(eng.ctx.lib.msx_last_error(engines[i % nprob].ctx.h).decode())

    # The collective: ONE RCCL all-gather of n float64 per rank per step.  Two routes to the same RCCL:
    #   msx_comm (default)  the library's own communicator (msx_comm_init: ncclCommInitRank from the copy of RCCL torch has
    #                       mapped; id broadcast through torch.distributed), ncclAllGather on the communicator's own stream,
    #                       forked from / joined to the launch stream by events -- plain HIP + RCCL calls, so the step loop
    #                       (kernel -> all-gather, double-buffered) can be captured into a hipGraph;
    #   torch.distributed   (MSX_BENCH_COLLECTIVE=torch, or when the communicator cannot be built) c10d's
    #                       all_gather_into_tensor, EAGER ONLY: c10d's watchdog thread polls the events of the works it
    #                       tracks, and a work recorded while a stream captures makes that poll abort the process ("operation
    #                       not permitted on an event last recorded in a capturing stream" -- seen here under
    #                       torch.distributed.run on the one-GPU rehearsal, intermittently).  Never captured.
    collective = 'none'
    h = eng.ctx.h
    if use_gather:
        collective = 'torch.distributed (eager)'
        if os.environ.get('MSX_BENCH_COLLECTIVE', 'rccl') == 'rccl':
            try:
                idt = torch.zeros(128, dtype=torch.uint8, device=dev)
                if rank == 0:
                    idt.copy_(torch.frombuffer(bytearray(eng.ctx.comm_unique_id()), dtype=torch.uint8))
                dist.broadcast(idt, src=0)
                eng.ctx.comm_init(bytes(idt.cpu().numpy().tobytes()), rank, world)
                collective = 'msx_comm (direct RCCL)'
            except Exception as exc:  # noqa: BLE001 - any setup problem: use the torch collective instead
                print('[bench] direct RCCL communicator unavailable ({}); using torch.distributed'.format(exc),
                      file=sys.stderr, flush=True)
        ok_t = torch.tensor([1 if collective.startswith('msx_comm') else 0], device=dev)
        dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)  # every rank must agree on the path
        if int(ok_t.item()) == 0:
            collective = 'torch.distributed (eager)'
    direct = collective.startswith('msx_comm')
    ag = eng.ctx.lib.msx_comm_allgather_dev
    wt = eng.ctx.lib.msx_comm_wait_slot
    pending = [False, False]
    overlap = not args.no_overlap   # the all-gather of step i next to the kernel of step i + 1 (else: one dependent chain)

    def gather(i, sp=sptr):
        b = i & 1
        if direct:
            if ag(h, C.c_void_p(logp[b].data_ptr()), C.c_void_p(gathered[b].data_ptr()), n, C.c_void_p(sp), b) != 0:
                raise RuntimeError(eng.ctx.lib.msx_last_error(h).decode())
            if not overlap:
                wt(h, b, C.c_void_p(sp))
            else:
                pending[b] = True
            return
        w = dist.all_gather_into_tensor(gathered[b], logp[b], async_op=True)
        if not overlap:
            w.wait()
        else:
            works[b] = w

    def reuse_guard(i, sp=sptr):
        b = i & 1
        if direct:
            if pending[b]:  # stream-level wait (no host block): step i-2's all-gather read logp[b]
                wt(h, b, C.c_void_p(sp))
                pending[b] = False
        elif works[b] is not None:
            works[b].wait()
            works[b] = None

    def drain(sp=sptr):
        for b in range(2):
            if direct and pending[b]:
                wt(h, b, C.c_void_p(sp))
                pending[b] = False
            if works[b] is not None:
                works[b].wait()
                works[b] = None

    # Device warm-up (setup, not the W warm-up steps of the contract): ~25 ms of the same launches, without the
    # collective, so that the timed region starts on a GPU in its sustained-load state.  After an idle period the
    # first ~10 ms of work run ~6 % slower (a 20-step run measured 18.5 us per kernel cold, 17.6 us after 1000
    # launches); the default 200-step run is long enough not to care, the driver's 20-step run is not.
    # (issued right before the timed region, after the graph capture: capturing leaves the GPU idle for milliseconds)
    ramp = int(os.environ.get('MSX_BENCH_RAMP', '1500'))
    # The step is launch-bound on the host next to a ~15 us kernel, so a run of `chunk` steps (kernel -> all-gather,
    # double-buffered exactly as above) is captured into one hipGraph and replayed; the timed region still executes
    # exactly K steps (K // chunk replays, the remainder eagerly).  Capture only is inside the try: the ranks FIRST agree
    # whether every one of them captured, and only then does anybody replay (a replay holds `chunk` collectives: a rank
    # that failed to capture must not meet it with an all-reduce).  MSX_BENCH_GRAPH=0 gives the eager loop;
    # MSX_BENCH_FAIL_CAPTURE_RANK=r makes rank r fail on purpose; the ordering itself is
    # mcmc_spec_amd.benchutil.capture_agreed (tests/test_dist_gloo.py).
    # The captured run covers as much of the timed region as possible: one replay costs the host ~10-16 us (the
    # guide's graph-replay floor), which a 20-step run (what the driver times) would otherwise pay several times.
    # Step i of a replay uses problem i mod nprob, theta batch i mod nbatch, output buffer i mod 2: any even chunk.
    # (with a collective in the loop only the direct-RCCL route is captured: see above)
    graph, chunk, head = None, 0, 0
    want_graph = os.environ.get('MSX_BENCH_GRAPH', '1') == '1'
    can_graph = want_graph and (direct or not use_gather) and args.steps >= 4
    from mcmc_spec_amd.benchutil import capture_agreed
    if can_graph:
        # (kernel-only runs: the first four steps go out as plain launches, so that the GPU is already busy while the
        # host prepares the graph launch -- 25-40 us that a 20-step run would otherwise spend with the GPU idle; same-box
        # 20-step runs this round: 15.4-16.0 us per step with four, 16.2-16.9 with none, 16.4-16.6 all eager)
        head = int(os.environ.get('MSX_BENCH_HEAD', '4')) if (not use_gather and args.steps >= 8 and os.environ.get('MSX_BENCH_NO_HEAD') != '1') else 0
        head = max(0, min(head, args.steps - 4))
        chunk = min(args.steps - head, int(os.environ.get('MSX_BENCH_GRAPH_CHUNK', '200'))) // 2 * 2

    def do_capture():
        if os.environ.get('MSX_BENCH_FAIL_CAPTURE_RANK') == str(rank):
            raise RuntimeError('capture failure requested for this rank')
        torch.cuda.synchronize(dev)
        g_ = torch.cuda.CUDAGraph()
        # thread_local: calls made by other threads (c10d's watchdog) must not invalidate the capture
        with torch.cuda.graph(g_, stream=torch.cuda.Stream(dev), capture_error_mode='thread_local'):
            cs = torch.cuda.current_stream(dev).cuda_stream
            tab = calls_for(cs)
            for i in range(chunk):
                reuse_guard(i, cs)
                launch(i, tab)
                if use_gather:
                    gather(i, cs)
            drain(cs)   # (every fork of the capture -- the communicator's stream -- is joined again)
        return g_

    def do_replay(g_):
        g_.replay()
        torch.cuda.synchronize(dev)

    def all_min(flag):
        if not (world > 1 or force_gather):
            return flag
        ok_t = torch.tensor([flag], device=dev)
        dist.all_reduce(ok_t, op=dist.ReduceOp.MIN)
        return int(ok_t.item())

    # N > 1: which kernel variant next to the collective, and the collective NEXT TO the following launch or BEHIND it?
    #  * The N = 1 variant needs every CU for itself (123 KB of LDS each): if RCCL's kernel cannot share a CU with it, a
    #    displaced walker costs a second round; the <= 128-VGPR variant leaves room but is the slower kernel (no LDS for the
    #    staged pixel vectors: DESIGN.md section 4).
    #  * Overlapping the all-gather of step i with the launch of step i + 1 hides the collective's latency -- unless its
    #    kernel displaces a walker (one-GPU rehearsal, one-rank communicator: 23.5 us per step overlapped, 16.6 as one
    #    dependent chain kernel -> all-gather -> kernel; across eight GPUs the collective takes longer and the balance
    #    may tip the other way).
    # Neither can be known from here, so the candidates are TIMED during set-up -- each as the very loop that follows
    # (its hipGraph where the loop is captured, two replays; else 40 eager steps), collective included -- and the ranks
    # agree on the one whose SLOWEST rank is fastest.  A candidate with a walker error status on any rank is out whatever
    # its time: a launch that fails is the fastest one (a timed-out linked form returns MSX_W_HANDOVER after one load).
    block_tuned = None
    if use_gather and not args.block and os.environ.get('MSX_BENCH_TUNE_BLOCK', '1') == '1':
        if args.npix < 8192:
            cands = [('own CU (pixel vectors staged in LDS), all-gather behind the launch', 0, _lib.PATH_AUTO, False),
                     ('own CU (pixel vectors staged in LDS), all-gather next to the following launch', 0, _lib.PATH_AUTO, True),
                     ('shared (<= 128 VGPRs, two per CU), all-gather next to the following launch', _lib.BLOCK_512_SHARED, _lib.PATH_AUTO, True)]
        else:  # (long spectra: the LINKED form's workgroups fill every CU and wait for each other; the fused kernel leaves half the CUs free)
            cands = [('automatic (linked while walkers x segments <= #CUs), all-gather behind the launch', 0, _lib.PATH_AUTO, False),
                     ('automatic (linked while walkers x segments <= #CUs), all-gather next to the following launch', 0, _lib.PATH_AUTO, True),
                     ('fused (one workgroup per walker), all-gather next to the following launch', 0, _lib.PATH_FUSED, True)]
        if args.no_overlap:
            cands = [c_ for c_ in cands if not c_[3]]
        tms, errs, graphs = [], [], []
        for _, cb, cpath, cov in cands:
            block, overlap = cb, cov
            for e in engines:
                e.ctx.set_path(cpath)
            calls = calls_for(sptr)
            for i in range(8):
                reuse_guard(i)
                launch(i)
                gather(i)
            drain()
            torch.cuda.synchronize(dev)
            for b_ in range(2):
                status[b_].zero_()
            g_c = capture_agreed(do_capture, do_replay, all_min, rank) if can_graph else None
            if can_graph and g_c is None:
                works[0] = works[1] = None
                pending[0] = pending[1] = False
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(stream)
            if g_c is not None:
                g_c.replay()
                g_c.replay()
                nst = 2 * chunk
            else:
                nst = 40
                for i in range(nst):
                    reuse_guard(i)
                    launch(i)
                    gather(i)
                drain()
            e1.record(stream)
            torch.cuda.synchronize(dev)
            tms.append(e0.elapsed_time(e1) * 1e3 / nst)
            errs.append(float((status[0] > _lib.W_REJECT).sum().item() + (status[1] > _lib.W_REJECT).sum().item()))
            graphs.append(g_c)
        t_all = torch.tensor(tms, dtype=torch.float64, device=dev)
        e_all = torch.tensor(errs, dtype=torch.float64, device=dev)
        dist.all_reduce(t_all, op=dist.ReduceOp.MAX)
        dist.all_reduce(e_all, op=dist.ReduceOp.MAX)
        t_rank = torch.where(e_all > 0, torch.full_like(t_all, float('inf')), t_all)
        pick = int(torch.argmin(t_rank).item())
        block, overlap = cands[pick][1], cands[pick][3]
        for e in engines:
            e.ctx.set_path(cands[pick][2])
        calls = calls_for(sptr)
        graph = graphs[pick]
        for b_ in range(2):
            status[b_].zero_()
        block_tuned = {'candidates_us_per_step': {cands[k][0]: float(t_all[k].item()) for k in range(len(cands))},
                       'candidates_walker_errors': {cands[k][0]: int(e_all[k].item()) for k in range(len(cands))},
                       'timed_as': 'hipGraph replays' if graph is not None else 'eager loop', 'taken': cands[pick][0]}
    for i in range(args.warmup):
        reuse_guard(i)
        launch(i)
        if use_gather:
            gather(i)
    drain()
    if can_graph and block_tuned is None:
        graph = capture_agreed(do_capture, do_replay, all_min, rank)
        if graph is None:
            works[0] = works[1] = None
            pending[0] = pending[1] = False
    # HIP events on the launch stream bracket runs of `ev_run` consecutive launches inside the timed region
    # (an event pair around every single launch would put two extra packets between back-to-back kernels
    # and inflate what it measures); kernel_ms = elapsed / ev_run, i.e. duration + the stream's launch gap
    ev_run = 8
    nev = args.steps // ev_run
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(max(nev, 1))]
    # the device warm-up (see above): at least `ramp` launches, then on in runs of 500 until two consecutive runs take
    # the same time to 0.7 % (a box that sat idle needs more than one that has just run the tests), 20,000 at most
    ramp_done = 0
    if ramp > 0:
        ra, rb = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        prev, stable = None, 0
        while ramp_done < ramp or (stable < 2 and ramp_done < 20000):
            ra.record(stream)
            for i in range(500):
                reuse_guard(i)
                launch(i)
            rb.record(stream)
            rb.synchronize()
            ramp_done += 500
            cur = ra.elapsed_time(rb)
            stable = stable + 1 if (prev is not None and abs(cur - prev) <= 0.007 * prev) else 0
            prev = cur
    gev_pool = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(8)]

    def timed_region():
        """EXACTLY K steps between barrier + synchronize on both sides; returns (seconds, graph-replay event pairs)."""
        nonlocal nev
        drain()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        first_eager = 0
        gev_ = []
        if graph is not None:
            for i in range(head):
                reuse_guard(i)
                launch(i)
            for _ in range((args.steps - head) // chunk):
                if not use_gather:  # kernels only in the graph: a replay's elapsed time / chunk is the kernel time
                    gev_.append(gev_pool[len(gev_)] if len(gev_) < len(gev_pool) else (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)))
                    gev_[-1][0].record(stream)
                graph.replay()
                if not use_gather:
                    gev_[-1][1].record(stream)
            first_eager = head + (args.steps - head) // chunk * chunk
            nev = 0  # (N > 1: kernel timed separately below)
        for i in range(first_eager, args.steps):
            reuse_guard(i)
            g, k = divmod(i, ev_run)
            if k == 0 and g < nev:
                ev[g][0].record(stream)
            launch(i)
            if use_gather:
                gather(i)
            if k == ev_run - 1 and g < nev:
                ev[g][1].record(stream)
        drain()
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)
        return time.perf_counter() - t0, gev_

    dt, gev = timed_region()
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    gather_ok = None
    if use_gather:  # the gathered vectors of the last two steps: own slice == own output, and == a fresh all-gather
        ok = all(torch.equal(gathered[b][rank * n:(rank + 1) * n], logp[b]) for b in range(2))
        ref = torch.empty_like(gathered[0])
        dist.all_gather_into_tensor(ref, logp[0])
        ok = ok and torch.equal(ref, gathered[0])
        okt = torch.tensor([1 if ok else 0], device=dev)
        dist.all_reduce(okt, op=dist.ReduceOp.MIN)
        gather_ok = bool(int(okt.item()))
    if gev:
        kern_ms = float(np.mean([a.elapsed_time(b) for a, b in gev])) / chunk
        kern_samples = len(gev) * chunk
    elif nev > 0 and world == 1 and graph is None:
        kern_ms = float(np.mean([a.elapsed_time(b) for a, b in ev[:nev]])) / ev_run
        kern_samples = nev * ev_run
    else:  # N > 1: collectives share the stream timeline; time the kernel alone after the timed region
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for i in range(ev_run):
            launch(i)
        e1.record(stream)
        torch.cuda.synchronize(dev)
        kern_ms = e0.elapsed_time(e1) / ev_run
        kern_samples = ev_run
    bad_t = ((status[0] > _lib.W_REJECT).sum() + (status[1] > _lib.W_REJECT).sum()).to(torch.int64).reshape(1)
    if world > 1:
        dist.all_reduce(bad_t, op=dist.ReduceOp.SUM)   # every rank's walkers count, and every rank leaves with the same code
    bad = int(bad_t.item())
    n_ranks_seen = dist.get_world_size() if (world > 1 or force_gather) else 1
    # ---- beside the headline: the SAME K steps without the device warm-up --------------------------------------
    # (N = 1 only.)  The GPU is left idle for a second, the W warm-up steps are repeated and the timed region runs
    # again, with no untimed launches in between: what a caller gets who evaluates one short burst now and then.
    unramped = None
    if world == 1 and not use_gather and os.environ.get('MSX_BENCH_NO_COLD') != '1':
        time.sleep(1.0)
        for i in range(args.warmup):
            launch(i)
        dt_cold, _ = timed_region()
        unramped = {'ms_per_step': dt_cold / args.steps * 1e3, 'value': n * world * args.steps / dt_cold,
                    'note': 'same {} steps after 1 s of idle + the {} warm-up steps, no untimed launches before the '
                            'timed region'.format(args.steps, args.warmup)}
    # ---- beside the headline: the clock the CUs ran at under this kernel's load ------------------------------------
    # (rank 0; untimed.)  Boxes of one pool differ: the same build has read 14.5 and 16.6 us per kernel in 20-step runs on
    # two leases of one afternoon.  A few launches with clock stamps (msx_probe_launch: thread 0 of every walker's
    # workgroup reads the 100 MHz wall clock and the shader-cycle counter at its first and last line), issued behind a
    # burst of plain launches so that the GPU is in its sustained-load state, say whether that is the clock.
    clock_probe = None
    if rank == 0 and not replicas and os.environ.get('MSX_BENCH_NO_PROBE') != '1':
        try:
            pr = []
            for k in range(5):
                for i in range(200):
                    launch(i)
                pr.append(engines[0].ctx.probe_launch(thetas[0].data_ptr(), n, ndim, logp[0].data_ptr(), status[0].data_ptr(), sptr,
                                                      _lib.MODE_LOGPOST, block))
            clock_probe = {k_: float(np.median([p_[k_] for p_ in pr])) for k_ in pr[0]}
            clock_probe['note'] = ('median of 5 probe launches, each behind 200 plain launches: shader clock while the walkers ran '
                                   '(cycle counter / 100 MHz wall clock), a walker\'s own time first line -> last line, and first '
                                   'walker\'s start -> last walker\'s end inside one launch')
        except Exception as exc:  # noqa: BLE001 - a diagnostic, never fatal
            clock_probe = {'error': str(exc)}
    # ---- N > 1: where a step's time goes (an untimed, eager pass after the timed region) ------------------------
    diag = None
    if use_gather:
        nd_ = 64
        t_launch = t_gather = t_wait = 0.0
        torch.cuda.synchronize(dev)
        d0, d1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        d0.record(stream)
        for i in range(nd_):
            a = time.perf_counter()
            reuse_guard(i)
            b = time.perf_counter()
            launch(i)
            c_ = time.perf_counter()
            gather(i)
            d_ = time.perf_counter()
            t_wait += b - a
            t_launch += c_ - b
            t_gather += d_ - c_
        drain()
        d1.record(stream)
        torch.cuda.synchronize(dev)
        diag = {'steps': nd_, 'host_us_per_step': {'wait_for_buffer': t_wait / nd_ * 1e6, 'kernel_issue': t_launch / nd_ * 1e6,
                                                   'collective_issue': t_gather / nd_ * 1e6},
                'stream_us_per_step_eager': d0.elapsed_time(d1) / nd_ * 1e3, 'kernel_alone_us': kern_ms * 1e3,
                'note': 'eager loop (kernel -> all-gather, double-buffered), this rank; the timed region itself ran as: '
                        + ('hipGraph replay' if graph is not None else 'eager loop')}

    if rank == 0:
        kern_s = kern_ms * 1e-3
        # the variant THIS launch took and the bytes one walker's workgroup requests from the memory system (L2-served):
        # asked of the library's own launcher (msx_launch_info reads the table the launcher reads -- nothing is mirrored here)
        info = eng.ctx.launch_info(n, _lib.MODE_LOGPOST, block)
        req = info['requested_bytes_per_eval']
        kernel_name = info['kernel']
        # ---- the roofline that bounds THIS design --------------------------------------------------------------
        # The kernel never streams the windowed grid from HBM: staging folds the resample into per-node tables of
        # 12 bytes per pixel (R float64 + H float32, 5.1 MB at config 2) that live in L2 / Infinity Cache, and a
        # walker's workgroup pulls its eight rows through its CU's memory pipeline.  The bound is the L2 -> CU path: at
        # 256 walkers one phase of the kernel (the blend, ~45 % of it) runs at that limit and the rest is a latency
        # chain; at large batches the WHOLE kernel averages 0.6 of the L2 aggregate (the sweep's last rows: above the
        # guide's own measured gather rate) with the vector ALUs half busy (`valu`, the sweep's valu_issue_frac).  The
        # contract's HBM figure is kept, labelled, in `hbm_contract`; it is NOT a bound on this design and exceeds 1.
        achieved = n * req / kern_s / 1e9
        traffic, traffic_src = None, None
        tj = load_profile('r3_logprob_traffic.json') or load_profile('r2_logprob_traffic.json')
        if tj and tj.get('config') == {'walkers': n, 'npix': args.npix, 'phot': bool(args.phot)}:
            traffic = tj.get('hbm_bytes_per_launch')
            traffic_src = 'profiles/ (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, separate passes, per launch)'
        roofline = {
            'bound': 'l2->cu', 'achieved': achieved, 'peak': L2_PEAK_GBPS, 'unit': 'GB/s', 'frac': achieved / L2_PEAK_GBPS,
            'traffic': traffic, 'traffic_source': traffic_src,
            'kernel': kernel_name, 'kernel_ms': kern_ms, 'kernel_ms_samples': kern_samples,
            'kernel_resources': {k_: info[k_] for k_ in ('form', 'threads', 'vgprs', 'static_lds_bytes', 'dynamic_lds_bytes', 'workgroups')},
            'requested_bytes_per_eval': req, 'requested_bytes_per_launch': n * req,
            'per_cu_GBps': req / kern_s / 1e9 if n <= 256 else None,
            'guide_l2_row_gather_GBps': {'per_cu': [66, 73], 'chip': [16800, 18800]},
            'frac_of_guide_measured_gather': achieved / 17800.0,
            'note': 'achieved = bytes the launch requests from the memory system (L2-served) / kernel time, averaged over '
                    'the WHOLE kernel (recipe, blend, median, chi^2); peak = the L2 aggregate of MI355X_MICROARCH.md, beside '
                    'it the same guide\'s measured chip-wide rate for rows gathered from L2 (16.8-18.8 TB/s).  The blend '
                    'phase alone moves its ~440 KB per walker in ~5.4 us: ~21 TB/s chip-wide with 256 CUs pulling, 81 GB/s '
                    'per CU (the guide measures 66-73); u and the data flux (64 KB) come in while the recipe runs; the rest of '
                    'the kernel is the walker\'s dependent chain (recipe, median, chi^2) and moves little.  Large batches '
                    'reach 16 TB/s over the whole kernel with half the requests per walker (pair form, extra.sweep)',
        }
        if not replicas:
            nwin = W['nwin']
            b_alg = 2 * 4 * nwin * 8 + ndim * 8 + 8  # SURVEY.md §8(d), float64 grid
            alg = n * b_alg / kern_s / 1e9
            roofline['hbm_contract'] = {
                'algorithmic_bytes_per_eval': b_alg, 'algorithmic_bytes_per_launch': n * b_alg, 'achieved_GBps': alg,
                'peak_GBps': HBM_PEAK_GBPS, 'ratio_to_hbm_peak': alg / HBM_PEAK_GBPS,
                'hbm_traffic_frac_of_peak': (traffic / kern_s / 1e9 / HBM_PEAK_GBPS) if traffic else None,
                'note': 'SURVEY §8(d) contract figure: B_alg = nspec*4*Nwin*8 + ndim*8 + 8 of the windowed float64 grid. '
                        'The kernel does not move these bytes (pair table, L2-resident), so this ratio is not a '
                        'fraction of a bound and exceeds 1; measured HBM traffic is the figure beside it'}
        copy_gbps = eng.ctx.stream_copy_gbps(int(args.copy_gib * (1 << 30)), 10)
        roofline['measured_stream_copy_GBps'] = copy_gbps
        vj = load_profile('r4_valu.json') or load_profile('r3_valu.json')
        vj_name = 'profiles/r4_valu.json' if load_profile('r4_valu.json') else 'profiles/r3_valu.json'
        issue_peak = 1024 * 2.4e9 / 4.0   # FP64 wave-instructions per second: 1024 SIMDs, one per 4 clocks, 2.4 GHz
        valu_by_regime = {}
        if vj:
            for pt in vj['points']:   # (keyed by the form that RAN at the profiled point: 'fused' / 'pair' / 'linked')
                valu_by_regime[(pt['npix'], pt['walkers'], pt.get('path', 'fused'))] = pt['valu_insts_per_eval']
            v = valu_by_regime.get((args.npix, n, info['form'].split(' ')[0]))
            if v:
                roofline['valu'] = {'insts_per_eval': v, 'wave_insts_per_s': v * n / kern_s, 'issue_peak_per_s': issue_peak,
                                    'issue_frac': v * n / kern_s / issue_peak,
                                    'source': vj_name + ' (rocprofv3 --pmc SQ_INSTS_VALU) x this run\'s kernel time; '
                                              'peak = 1024 SIMDs x 2.4 GHz / 4 clocks per FP64 wave-instruction'}
        workload = ('BASELINE config 5: KOI targets ({} px each after the (0.55, 0.90) um crop), one independent problem per '
                    'GPU, {} walkers per launch, logposterior with 2 contrast terms; {}'.format(
                        args.npix, n, 'ONE GPU rehearsal: 8 staged problems resident, launched round-robin' if world == 1
                        else 'replicas, no collective')) if replicas else (
            'binary (T1=3850/T2=3025) {}-pixel spectrum{}, {} walkers per GPU per launch, logposterior (prior gate + '
            'likelihood){}'.format(args.npix, ' + 6-band photometry' if args.phot else ' + 2 contrast terms', n,
                                   (', RCCL all-gather of log-probs' + (' overlapped with the next launch' if overlap else
                                                                      ' before the next launch'))
                                   if world > 1 else ''))
        out = {
            'metric': 'walker log-likelihood evals/sec (whole node)',
            'value': n * world * args.steps / dt, 'unit': 'evals/s', 'n_gpus': world, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None, 'data': 'synthetic',
            'dtype': 'f64' if args.store == 'f64' else 'f64 arithmetic over a float32-STORED grid table (--store f32: a separately labelled precision, not the headline)',
            # (what the arithmetic type does not say: two of the staged tables only ever enter multiplied by eps ~ 4e-5 A_V and
            # are stored in float32 -- csrc/blend.h; every sum, the model vector and the grid's R table are float64)
            'storage': {'R': args.store, 'H': 'f32', 'dk': 'f32', 'note': 'R = lo + (hi - lo) t per grid node and pixel; H = hi t and dk = k_hi - k_lo enter the model only as eps * H, eps = 2^(redc dk) - 1: their 2^-24 rounding perturbs a pixel by < 2.4e-12 A_V relative (tests/test_gpu_parity.py, A_V = 3)'},
            'config': {'workload': workload, 'baseline_config': args.config if args.config != 2 or world == 1 else 3,
                       'walkers_total': n * world, 'npix': args.npix, 'nwin': W.get('nwin'),
                       'grid': '26x4x135000 f64 synthetic',
                       'block_threads': block or 'auto', 'block_tuned_next_to_the_collective': block_tuned, 'collective': collective,
                       'device_warmup': '{} untimed launches (>= {}, until two runs of 500 agree to 0.7 %) after the {} warm-up steps, right before the timed region (sustained-load clocks)'.format(ramp_done, ramp, args.warmup),
                       'step_loop': ('{} eager + hipGraph of {} steps x {} replays + {} eager'.format(
                           head, chunk, (args.steps - head) // chunk, args.steps - head - (args.steps - head) // chunk * chunk)
                           if graph is not None else 'eager')},
            'roofline': roofline,
            'walker_error_statuses': bad, 'gather_verified': gather_ok, 'n_ranks_seen': n_ranks_seen,
        }
        if bad > 0 or gather_ok is False:
            # evaluations that failed are not evaluations: no headline from them (and a non-zero exit code below)
            out['invalid'] = ('{} walker error statuses in the timed region\'s last two steps'.format(bad) if bad > 0
                              else 'the gathered log-probabilities differ from the ranks\' own')
            out['value_if_it_were_valid'] = out['value']
            out['value'] = None
        if unramped is not None:
            out['unramped'] = unramped
        if clock_probe is not None:
            out['clock_probe'] = clock_probe
        if diag is not None:
            out['multi_gpu_diag'] = diag
        out['cpu_baseline'] = None
        extra = {}
        if world == 1 and not replicas and not args.no_extras and args.npix == 4096 and not args.phot:
            # what an emcee run gets: ONE dependent chain of this ensemble, walker state resident on the GPU
            # (DeviceEnsembleSampler; consecutive half-steps overlap on two streams).  Before the CPU
            # baseline: the chain's launches are issued by this host thread, and the baseline's pools leave the host busy
            from mcmc_spec_amd.sampler import DeviceEnsembleSampler
            p0 = synth.draw_walkers(args.walkers, seed=9, tmin=W['tmin'], tmax=W['tmax'])
            nit = 3000
            chain = {}
            # 'device': the library draws the move's randomness on the GPU (counter-based generator, one launch per chunk,
            # nothing uploaded); 'host': NumPy draws it, two threads, pipelined one chunk ahead -- the stream the host loop shares
            for rng_mode in ('device', 'host'):
                smp = DeviceEnsembleSampler(args.walkers, p0.shape[1], eng, seed=1, chunk=100, rng=rng_mode)
                smp.run_mcmc(p0, 400, store=False)   # (clocks, the chunk sizes' ramp, the host's random-number threads)
                t0 = time.perf_counter()
                smp.run_mcmc(p0, nit, store=False)
                dt_c = time.perf_counter() - t0
                chain[rng_mode] = {'us_per_iteration': dt_c / nit * 1e6, 'evals_per_s': args.walkers * nit / dt_c,
                                   'acceptance': float(smp.acceptance_fraction.mean()), 'overlapped_half_steps': bool(smp.overlapped)}
            extra['dependent_chain'] = dict(chain['device'], walkers=args.walkers, iterations=nit, randomness='device (counter-based generator)',
                                            host_randomness=chain['host'],
                                            note='wall time of run_mcmc (chain download included); consecutive half-steps overlap on two '
                                                 'streams, a walker is handed over as tagged 8-byte words (include/msx.h)')
        if want_cpu:
            th_cpu = synth.draw_walkers(8192, seed=77, tmin=W['tmin'], tmax=W['tmax'])
            g, st = eng.ctx.logprob_batch(th_cpu, _lib.MODE_LOGPOST)
            procs = (0,) if args.no_extras else (0, 15)   # 15 = the reference's hard-coded pool width (mft6.py:1744)
            recs = cpu_baseline(W, th_cpu, g, args.cpu_budget, procs)
            out['cpu_baseline'] = recs[0]
            if len(recs) > 1:
                extra['cpu_baseline_pool15'] = recs[1]
        if world == 1 and not replicas and not args.no_extras:
            # batch-size sweep on the staged problem (device time per launch, automatic variant choice)
            sweep = []
            # (the CPU baseline above left the GPU idle for tens of seconds: the clocks are brought back up first -- ~50 ms
            # of launches, untimed -- and every row is the median of three timed blocks behind ten untimed launches)
            th_w = torch.from_numpy(synth.draw_walkers(1024, seed=3, tmin=W['tmin'], tmax=W['tmax'])).to(dev)
            device_time_us(eng, th_w, torch.empty(1024, dtype=torch.float64, device=dev), torch.empty(1024, dtype=torch.int32, device=dev),
                           stream, 1024, 1500)
            for m in (128, 256, 512, 1024, 2048, 4096, 16384):
                th = torch.from_numpy(synth.draw_walkers(m, seed=3, tmin=W['tmin'], tmax=W['tmax'])).to(dev)
                lp_, st_ = torch.empty(m, dtype=torch.float64, device=dev), torch.empty(m, dtype=torch.int32, device=dev)
                us = device_time_us(eng, th, lp_, st_, stream, m, max(30, min(60, 400000 // m)), warm=10, reps=3)
                # the form the launches TOOK (msx_last_form: MSX_PATH_AUTO looks at the planner's counts) and what that form
                # requests per walker -- from the library, like the headline's
                ran = eng.ctx.last_form()
                inf_m = eng.ctx.launch_info(m, _lib.MODE_LOGPOST, 0)
                req_m = inf_m['requested_bytes_per_eval'] if inf_m['form_id'] == ran else None
                row = {'walkers': m, 'device_us': us, 'evals_per_s': m / us * 1e6, 'form': _lib.FORM_NAMES[ran],
                       'kernel': inf_m['kernel'].split(' (')[0] if inf_m['form_id'] == ran else None, 'requested_bytes_per_eval': req_m}
                if req_m:
                    row.update(requested_GBps=m * req_m / us / 1e3, frac_of_l2_peak=m * req_m / us / 1e3 / L2_PEAK_GBPS)
                # VALU wave-instructions per evaluation of the nearest profiled point of the same form (profiles/r4_valu.json)
                cands_v = [(abs(np.log(k_[1] / m)), v_) for k_, v_ in valu_by_regime.items() if k_[0] == args.npix and k_[2] == _lib.FORM_NAMES[ran].split(' ')[0]]
                if cands_v and min(cands_v)[0] < 0.75:
                    vi = min(cands_v)[1]
                    row['valu_insts_per_eval'] = vi
                    row['valu_issue_frac'] = vi * m / (us * 1e-6) / issue_peak
                sweep.append(row)
            extra['sweep'] = {'npix': args.npix, 'rows': sweep}
            if args.npix == 4096 and not args.phot:
                # BASELINE config 4's per-GPU share: 16,384 px + 6-band photometry, 1,024 walkers / 8 GPUs
                e4 = Engine(local)
                W4 = build_workload(e4, 16384, True, grid=W.get('flux'))
                th = torch.from_numpy(synth.draw_walkers(128, seed=3, tmin=W4['tmin'], tmax=W4['tmax'])).to(dev)
                lp_, st_ = torch.empty(128, dtype=torch.float64, device=dev), torch.empty(128, dtype=torch.int32, device=dev)
                us = device_time_us(e4, th, lp_, st_, stream, 128, 50)
                inf4 = e4.ctx.launch_info(128, _lib.MODE_LOGPOST, 0)
                extra['config4_per_gpu_share'] = {'walkers': 128, 'npix': 16384, 'photometry_bands': 6, 'device_us': us,
                                                  'evals_per_s': 128 / us * 1e6, 'form': _lib.FORM_NAMES[e4.ctx.last_form()],
                                                  'kernel': inf4['kernel'].split(' (')[0],
                                                  'requested_bytes_per_eval': inf4['requested_bytes_per_eval']}
        if world == 1 and not replicas and not args.no_extras and args.npix == 4096 and not args.phot and args.store == 'f64':
            # A separately labelled precision, beside the headline and never instead of it: the same problem with the
            # grid table R STORED in float32 (include/msx.h, msx_set_grid_storage; the arithmetic stays float64)
            e32 = Engine(local)
            build_workload(e32, args.npix, False, grid=W.get('flux'), store='f32')
            th32 = synth.draw_walkers(n, seed=3, tmin=W['tmin'], tmax=W['tmax'])
            t32 = torch.from_numpy(th32).to(dev)
            lp_, st_ = torch.empty(n, dtype=torch.float64, device=dev), torch.empty(n, dtype=torch.int32, device=dev)
            us32 = device_time_us(e32, t32, lp_, st_, stream, n, 200)
            us64 = device_time_us(eng, t32, logp[0], status[0], stream, n, 200)
            g32, _ = e32.ctx.logprob_batch(th32, _lib.MODE_LOGPOST)
            g64, _ = eng.ctx.logprob_batch(th32, _lib.MODE_LOGPOST)
            fin = np.isfinite(g64)
            inf32 = e32.ctx.launch_info(n)
            extra['f32_stored_grid_table'] = {
                'walkers': n, 'device_us': us32, 'evals_per_s': n / us32 * 1e6, 'device_us_f64_same_loop': us64,
                'max_rel_dev_from_f64_tables': float(np.max(np.abs(g32[fin] - g64[fin]) / np.abs(g64[fin]))),
                'kernel': inf32['kernel'].split(' (')[0], 'requested_bytes_per_eval': inf32['requested_bytes_per_eval'],
                'note': 'labelled precision: R stored in float32 (2^-24 on the grid values), float64 arithmetic; plain launches back '
                        'to back, HIP events; BASELINE tolerance 1e-6 relative'}
            # the broadening placed INSIDE the evaluation (include/msx.h, MSX_PATH_INPATH): a form of its own,
            # never the automatic choice -- timed beside the default form in the same loop, values compared
            ein = Engine(local)
            build_workload(ein, args.npix, False, grid=W.get('flux'), broaden='in_path')
            ein.ctx.set_path(_lib.PATH_INPATH)
            us_in = device_time_us(ein, t32, lp_, st_, stream, n, 100)
            g_in, _ = ein.ctx.logprob_batch(th32, _lib.MODE_LOGPOST)
            extra['in_path_broadening'] = {
                'walkers': n, 'device_us': us_in, 'evals_per_s': n / us_in * 1e6, 'device_us_default_form_same_loop': us64,
                'max_rel_dev_from_default_placement': float(np.max(np.abs(g_in[fin] - g64[fin]) / np.abs(g64[fin]))),
                'kernel': ein.ctx.launch_info(n)['kernel'].split(' (')[0],
                'note': 'mft6.py:124-152 applied per walker to the unreddened composite inside the data window (the call the reference '
                        'keeps commented out at :550, placed as SURVEY A3 (ii)); the default -- and the headline -- broadens per grid '
                        'node at staging, mft6.py:366-378'}
        if extra:
            out['extra'] = extra
        sys.stdout.flush()
        os.write(result_fd, (json.dumps(out, default=lambda o: o.item() if hasattr(o, 'item') else str(o)) + '\n').encode())
    if world > 1 or force_gather:
        dist.destroy_process_group()
    if bad > 0 or gather_ok is False:
        raise SystemExit(3)


if __name__ == '__main__':
    main()
