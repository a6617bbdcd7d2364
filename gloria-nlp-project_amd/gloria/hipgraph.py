"""hipGraph safety on this ROCm stack (7.2, gfx950) - measured, round 3 (profiles/r03_hipgraph_race.txt).

A graph captured from the image encoder's backward replays WRONG from its second replay on: NaN / garbage weight
gradients in layer1 and conv1 while everything behind layer2 is right; correct under rocprofv3 (which serialises
kernels), correct with `DEBUG_CLR_GRAPH_PACKET_CAPTURE=0`, wrong with every MIOpen solver family and with torch's own
BatchNorm.  Cause: the captured graph holds MEMSET nodes (MIOpen's CK backward-data solver zero-fills dx with
hipMemsetAsync: nine per backward pass, `__amd_rocclr_fillBufferAligned`), and the HIP runtime's graph "packet capture"
fast path (kernel nodes replayed as pre-built AQL packets) does not order those memsets against the kernel packets
around them.  Whether a replay is wrong depends on timing (the same graphs passed three training steps next to a busy
second stream and failed alone on one stream), so a test that passes proves nothing.

Two defences, both used:
  * `DEBUG_CLR_GRAPH_PACKET_CAPTURE=0` is put in the environment before the HIP runtime starts (this module is imported
    first by the package, bench.py, tests/conftest.py and __graft_entry__.py; the runtime reads its flags at its first
    API call).  Graph nodes then go through the normal command path - stream-ordered, still no Python per launch.
    Without the flag ("1" exported by the user, or the GPU initialised before this module could set it) nothing is
    captured (`usable`).
  * every capture is VERIFIED before it is used (`verify_capture`): three replays on the capture's sample input against
    an eager pass of the same module with the same generator state; a mismatch disables the graph with a warning and the
    eager path runs.  This catches a wrong capture, NOT the race: with packet capture on, a capture that passed the three
    replays produced inf / NaN in the loop that followed (tests/test_gpu_streams.py keeps that loop as a regression test)."""

import os
import warnings

ENV = "DEBUG_CLR_GRAPH_PACKET_CAPTURE"


def _decide():
    """True if graphs may be used: the flag is "0" and was in the environment before torch initialised the GPU"""
    import sys
    pre = os.environ.get(ENV)
    if pre is not None:
        return pre == "0"
    os.environ[ENV] = "0"
    t = sys.modules.get("torch")
    return not (t is not None and t.cuda.is_initialized())      # a flag check: no HIP call


SAFE = _decide()


def usable(name):
    """gate of every capture: verification cannot see this race (a capture that passed three verified replays went wrong
    in the training loop), so without the runtime flag there are no graphs at all"""
    if not SAFE:
        warnings.warn(f"{ENV}=0 was not in effect when the GPU was initialised: the {name} runs eagerly (no hipGraph)")
    return SAFE


def consistent(name, reference, replays, rel_tol=5e-2):
    """reference / replays[i]: lists of tensors (outputs + gradients) of the same pass, eager and through the graph.
    True if every replay is finite and within rel_tol (relative Frobenius norm over all tensors) of the eager pass."""
    import torch
    den = sum(float(t.float().pow(2).sum()) for t in reference) ** 0.5
    if not (den > 0 and den == den and den != float("inf")):
        warnings.warn(f"hipGraph check of the {name}: the eager reference pass is not finite - graph not used")
        return False
    for k, rep in enumerate(replays):
        num = 0.0
        for a, b in zip(reference, rep):
            d = (a.float() - b.float())
            num += float(torch.nan_to_num(d, nan=float("inf")).pow(2).sum())
        rel = num ** 0.5 / den
        if not rel <= rel_tol:
            warnings.warn(f"hipGraph replay {k} of the {name} differs from the eager pass (relative distance {rel:.3g}; "
                          f"{ENV}={os.environ.get(ENV)!r}): the graph is NOT used, the eager path runs instead")
            return False
    return True


def verify_capture(name, eager, graphed, sample, before_replay=None, n_replays=3, rel_tol=5e-2):
    """eager / graphed: the module and its graphed callable; sample: the capture's input tensors.  Runs one eager pass and
    n_replays graph passes of  loss = sum(mean(out^2))  with the SAME CUDA generator state (dropout inside the module draws
    the same masks on both paths, models/rng.py) and compares outputs and parameter gradients (`consistent`).
    before_replay: called before each graph pass (the owner's per-replay duty, e.g. GraphRng.refresh)."""
    import torch
    dev = sample[0].device
    params = [p for p in eager.parameters() if p.requires_grad]
    state = torch.cuda.get_rng_state(dev)

    def run(fn, hook):
        torch.cuda.set_rng_state(state, dev)
        if hook is not None:
            hook()
        outs = fn(*sample)
        outs = outs if isinstance(outs, (tuple, list)) else (outs,)
        loss = sum(o.float().pow(2).mean() for o in outs)
        grads = torch.autograd.grad(loss, params, allow_unused=True)
        got = [o.detach().float().clone() for o in outs] + [g.detach().float().clone() for g in grads if g is not None]
        torch.cuda.synchronize(dev)
        return got

    # make_graphed_callables returns the module itself with `forward` rebound to the graphed call: the eager pass is the
    # class's own forward
    ref = run(lambda *a: type(eager).forward(eager, *a), None)
    reps = [run(graphed, before_replay) for _ in range(n_replays)]
    ok = len({len(r) for r in reps + [ref]}) == 1 and consistent(name, ref, reps, rel_tol)
    torch.cuda.set_rng_state(state, dev)
    return ok
