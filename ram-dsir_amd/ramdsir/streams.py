"""Lane streams that REALLY run beside each other.

HIP multiplexes a process's streams onto a small number of hardware queues (4 by default on this stack) and which queue a
stream gets depends on how many streams the process created before it (torch's pool, a process group, a DataLoader pin
thread ...).  Two lanes of the fused step on one hardware queue serialise: measured 5.7 -> 6.5 ms/step with the main and
the restoration lane aliased, 9-11 ms with main and the weight-gradient lane aliased (scripts/stream_alias.py,
scripts/ddp_diag.py).  Raising GPU_MAX_HW_QUEUES is not the answer: with more than four ACTIVE queues
the data-parallel step took 9-10 ms.  So the lanes are picked by measurement: a candidate stream is accepted only if a tiny
kernel on it completes while a long spin kernel occupies each stream it has to run beside."""
import sys

import torch

_SPIN = 6_000_000            # cycles of torch.cuda._sleep: a few ms


def runs_beside(busy, probe, scratch):
    """True when work submitted to `probe` does not wait for work running on `busy`.  Timed on the GPU: events around the
    spin kernel on `busy` and behind a tiny kernel on `probe` -- host wall-clock jitter (a busy host, another rank
    sharing the device) cannot turn an independent queue into an aliased one or the other way round."""
    torch.cuda.synchronize()
    e0, e1, ep = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    with torch.cuda.stream(busy):
        e0.record(busy)
        torch.cuda._sleep(_SPIN)
        e1.record(busy)
    with torch.cuda.stream(probe):
        scratch.add_(1.0)
        ep.record(probe)
    torch.cuda.synchronize()
    return e0.elapsed_time(ep) < 0.3 * e0.elapsed_time(e1)


def _beside_consistent(busy, probe, scratch):
    """Two probes that agree, or the majority of three."""
    a, b = runs_beside(busy, probe, scratch), runs_beside(busy, probe, scratch)
    return a if a == b else runs_beside(busy, probe, scratch)


verified = {}                # id(stream) -> True / False: what the last pick_lanes call measured (TrainStep.lanes_verified)


def pick_lanes(n, device, beside, tries=12):
    """n new streams on `device`, each measured to run beside every stream in `beside` and beside each other.  Falls back to
    unverified streams (with a warning) when the stack does not offer enough independent queues."""
    if n <= 0:
        return []
    try:
        return _pick(n, device, list(beside), tries)
    except Exception as e:                                  # a stack without torch.cuda._sleep, a probe that fails: lanes still work
        print('[ramdsir] lane-stream probing failed (%s: %s); using unverified streams' % (type(e).__name__, e), file=sys.stderr)
        out = [torch.cuda.Stream(device=device) for _ in range(n)]
        for st in out:
            verified[id(st)] = False
        return out


def _pick(n, device, beside, tries):
    scratch = torch.zeros(256, device=device)
    chosen, rejected = [], []
    for _ in range(tries):
        if len(chosen) == n:
            break
        cand = torch.cuda.Stream(device=device)
        # a stream's FIRST submission can go to whichever hardware queue is idle at that moment; only from the second one on
        # does it show the queue it keeps (scripts/pick_debug.py): warm it up, then ask (twice, a third time on disagreement)
        with torch.cuda.stream(cand):
            scratch.add_(1.0)
        torch.cuda.synchronize()
        if all(_beside_consistent(b, cand, scratch) for b in list(beside) + chosen):
            chosen.append(cand)
            verified[id(cand)] = True
        else:
            rejected.append(cand)
    if len(chosen) < n:
        print('[ramdsir] only %d of %d lane streams run concurrently with the main stream on this stack; '
              'the remaining lanes share a hardware queue (slower, still correct)' % (len(chosen), n), file=sys.stderr)
        for st in rejected[:n - len(chosen)]:
            verified[id(st)] = False
        chosen += rejected[:n - len(chosen)]
        while len(chosen) < n:
            st = torch.cuda.Stream(device=device)
            verified[id(st)] = False
            chosen.append(st)
    return chosen
