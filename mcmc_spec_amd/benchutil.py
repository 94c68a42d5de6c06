"""Small host-side helpers of bench.py that must behave the same on every rank (tested with gloo on CPU)."""
from __future__ import annotations

import sys


def capture_agreed(capture, replay, all_min, rank=0, label='hipGraph capture of the step loop'):
    """Capture on this rank, then let ALL ranks agree before anybody replays.

    ``capture()`` builds and returns this rank's graph (may raise); ``replay(graph)`` runs it once (it may contain
    collectives); ``all_min(flag) -> int`` is a MIN all-reduce of an int over the ranks (identity when there is
    one rank).  The order is the point: a replay holds collectives, so a rank whose capture failed must not meet
    another rank's replay with the agreement all-reduce -- every rank first reports, and only if every rank
    captured does every rank replay; otherwise every rank drops its graph.  Returns the graph or None."""
    graph = None
    try:
        graph = capture()
    except Exception as exc:  # noqa: BLE001 - any failure means "eager loop", on every rank
        print('[bench] {} failed on rank {} ({}); eager loop on every rank'.format(label, rank, exc), file=sys.stderr,
              flush=True)
        graph = None
    ok = int(all_min(1 if graph is not None else 0))
    if ok == 0:
        return None
    replay(graph)  # one untimed replay, after the ranks agreed
    return graph
