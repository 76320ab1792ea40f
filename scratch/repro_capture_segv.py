"""Root cause of the host segfault in hipStreamEndCapture (round-1 gpurun_out/segv.log, round-2 r2d_tests.log).

Hypothesis: an autograd graph from an EARLIER eager iteration that is still alive at capture time (a caller keeps the loss
tensor) keeps its AccumulateGrad nodes alive; those nodes are bound to the stream of that iteration (the default stream).  In
the captured backward the engine then accumulates on THAT stream behind an event recorded on the capturing stream: the default
stream is pulled into the capture and never joined -> CUDA reports cudaErrorStreamCaptureUnjoined, HIP faults in EndCapture.

    python scratch/repro_capture_segv.py stale        # eager step on the default stream, loss kept alive, then capture
    python scratch/repro_capture_segv.py clean        # same, loss deleted before the capture
    python scratch/repro_capture_segv.py samestream   # loss kept alive, but the eager step ran on the capture stream
Each mode runs in this process; run them as separate processes (a fault kills the interpreter)."""
import os, sys
sys.path.insert(0, os.getcwd())
import torch
from self_driving_model_amd import runtime
from self_driving_model_amd.hip import ops as hops
from self_driving_model_amd.models.experts import BDDDrivableExpert
from self_driving_model_amd.training import synthetic

mode = sys.argv[1]
dev = torch.device("cuda:0")
torch.manual_seed(0)
m = BDDDrivableExpert(3, pretrained_backbone=False).to(dev).train()
b = synthetic.bdd_drivable_batch(2, 128, 160, 3, dev, seed=1)
cap_stream = torch.cuda.Stream()


def fwd_bwd():
    for p in m.parameters():
        if p.grad is not None:
            p.grad.zero_()
    loss = hops.CrossEntropy2d.apply(m(b["image"]), b["mask"], 255)
    loss.backward()
    return loss


with runtime.precision(torch.float16):
    if mode == "samestream":
        cap_stream.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(cap_stream):
            kept = fwd_bwd()
            kept2 = fwd_bwd()
        torch.cuda.current_stream().wait_stream(cap_stream)
    else:
        kept = fwd_bwd()
        kept2 = fwd_bwd()
    if mode == "clean":
        del kept, kept2
    torch.cuda.synchronize()
    if mode == "probe":
        # can the hazard be SEEN before capturing?  A tiny backward through every parameter on the capture stream, outside any
        # capture, with warnings recorded: does torch's AccumulateGrad stream-mismatch warning fire, and every time?
        import warnings
        params = [p for p in m.parameters() if p.requires_grad]
        for attempt in range(3):
            if attempt == 2:
                del kept, kept2  # graphs gone: the probe must come back clean
            with warnings.catch_warnings(record=True) as rec:
                warnings.simplefilter("always")
                with torch.cuda.stream(cap_stream):
                    torch.stack([p.flatten()[0] * 0.0 for p in params]).sum().backward()
                torch.cuda.synchronize()
            hits = [w for w in rec if "AccumulateGrad node's stream" in str(w.message)]
            print("probe attempt", attempt, "warnings", len(rec), "stream-mismatch", len(hits), flush=True)
        sys.exit(0)
    print(mode, "eager done; capturing", flush=True)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=cap_stream):
        loss = fwd_bwd()
    print(mode, "capture ended", flush=True)
    g.replay()
    torch.cuda.synchronize()
    print(mode, "replay ok, loss", float(loss), flush=True)
