"""TEST-ONLY: route stllm_amd.hip's *training* bindings to the host-emulated build of the same kernel sources
(tests/hipemu): the ctypes argument lists of hip.py and the kernels' logic are then exercised on CPU tensors."""
import contextlib
import ctypes
import importlib.util
import os

HERE = os.path.dirname(os.path.abspath(__file__))


def _builder():
    spec = importlib.util.spec_from_file_location("build_emu", os.path.join(HERE, "hipemu", "build_emu.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


ON_DEVICE = os.environ.get("STLLM_TRAIN_KERNELS_ON_DEVICE") == "1"   # set by tests/test_train_gpu.py for its child process
DRY_RUN = os.environ.get("STLLM_TRAIN_KERNELS_ON_DEVICE") == "dry"    # the device-mode plumbing (proxy) over the emulated library


def available():
    return ON_DEVICE or os.path.exists(_builder().CLANG)


class _DeviceProxy:
    """stllm_amd.hip on the real GPU for tests written against CPU tensors: tensor arguments are copied to the device, the real
    entry point runs, every tensor argument is copied back (in-place results) and tensor results are returned on the CPU."""

    def __init__(self, hip, to_device=None, sync=None):
        self._hip = hip
        self._to_device = to_device or (lambda x: x.cuda())
        self._sync = sync

    def __getattr__(self, name):
        import torch
        f = getattr(self._hip, name)

        def call(*a, **k):
            moved = []

            def mv(x):
                if isinstance(x, torch.Tensor):
                    y = self._to_device(x)
                    moved.append((x, y))
                    return y
                if isinstance(x, (tuple, list)):
                    return type(x)(mv(e) for e in x)
                return x

            def back(r):
                if isinstance(r, torch.Tensor):
                    return r.cpu()
                if isinstance(r, (tuple, list)):
                    return type(r)(back(x) for x in r)
                return r
            r = f(*[mv(x) for x in a], **{kk: mv(v) for kk, v in k.items()})
            (self._sync or torch.cuda.synchronize)()
            for x, y in moved:
                x.copy_(y.cpu())
            return back(r)
        return call


@contextlib.contextmanager
def emulated():
    from stllm_amd import hip
    if ON_DEVICE:
        yield _DeviceProxy(hip)
        return
    L = hip._bind(ctypes.CDLL(_builder().build()), strict=False)
    import torch
    saved = (hip._lib, hip._train_bound, hip._stream, hip._req, hip.gemm_workspace)
    hip._lib, hip._train_bound = L, False
    hip._stream = lambda: None
    hip._req = lambda t, dtype=None, what="tensor": t
    hip.gemm_workspace = lambda device: torch.zeros(16, dtype=torch.uint8)
    try:
        yield _DeviceProxy(hip, to_device=lambda t: t.detach().clone(), sync=lambda: None) if DRY_RUN else hip
    finally:
        hip._lib, hip._train_bound, hip._stream, hip._req, hip.gemm_workspace = saved
