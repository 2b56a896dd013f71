"""hipGraph capture of a static-shape training step.

One TCB-PSP R101 step is ~1 400 kernel launches issued from Python (ctypes + autograd bookkeeping): the host needs
about as long to enqueue them as the GPU needs to run them.  Shapes, pointers and launch geometry are identical from
step to step (fixed crop, fixed batch), so the whole step - zero_grad, forward, fused loss, backward, gradient
all-reduce, SGD - is recorded ONCE into a hipGraph (torch.cuda.CUDAGraph drives hipStreamBeginCapture on ROCm; every
kernel of libvspw_hip.so is launched on torch's current stream, i.e. the capturing stream) and replayed with one
hipGraphLaunch per step.

What makes the step capturable:
  * all device memory comes from torch's caching allocator (graph-private pool during capture); the C ABI never
    allocates or synchronises;
  * the inputs live in static tensors that the caller refreshes (`copy_`) before a replay;
  * the SGD kernel reads its learning rates from a small device array (optim.SGD.set_lrs) and its per-parameter table
    is uploaded by a memcpy node from pinned memory, so the poly schedule needs no re-capture;
  * Dropout2d masks come from torch's graph-safe Philox generator (a fresh mask per replay).
"""
import torch


class GraphedStep(object):
    """Capture `fn()` after `warmup` eager calls on a side stream; `replay()` re-runs it.  `fn` must be free of host
    synchronisation (.item(), .cpu(), pageable copies) and must use only static input tensors; whatever it returns
    (tensors) stays valid and is overwritten by each replay."""

    def __init__(self, fn, warmup=2, stream=None):
        """stream: capture on this (non-default) stream instead of a fresh one.  Needed when gradient hooks were
        registered before the capture (distributed.GradReducer): register_post_accumulate_grad_hook creates the
        AccumulateGrad nodes, which stay bound to the stream that was current THEN; if that is not the capturing stream
        the hooks' collectives run outside the capture as far as ProcessGroupNCCL can tell, their work objects go to its
        watchdog thread, and the watchdog aborts the process on the first captured event it polls
        (hipErrorCapturedEvent - seen 3 runs of 4 with a 1-rank RCCL group).  bench.py / train_clip2 therefore make ONE
        side stream current before they wrap the model and hand it in here."""
        if not torch.cuda.is_available():
            raise RuntimeError("GraphedStep needs a GPU (hipGraph capture)")
        self.fn = fn
        side = stream if stream is not None else torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(max(int(warmup), 1)):
                fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        self.graph = torch.cuda.CUDAGraph()
        # capture on the stream the warm-up ran on: autograd's AccumulateGrad nodes remember the stream they were
        # created on, and a mismatch with the capturing stream would add cross-stream waits on a non-capturing stream
        # capture_error_mode "thread_local": with a process group alive, ProcessGroupNCCL's watchdog THREAD polls its
        # work events (hipEventQuery) at any time; under the default "global" mode that call is illegal while this
        # thread captures, the watchdog throws and the process aborts (seen on MI355X with a 1-rank RCCL group).
        with torch.cuda.graph(self.graph, stream=side, capture_error_mode="thread_local"):
            self.outputs = fn()
        self.stream = side

    def replay(self):
        self.graph.replay()
        # a replay rewrites parameters and BatchNorm running statistics through raw pointers: neither tensor._version
        # nor the Python-side generation counters move, so the folded conv+BN weights cached for inference and the
        # transposed weights of the data-gradient GEMMs would otherwise be reused stale by the next eager call
        from . import ops

        ops.invalidate_inference_cache()
        return self.outputs
