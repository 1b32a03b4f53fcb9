"""Device-resident inputs / outputs of the C-ABI (direct_mem_t = DIRECT_MEM_DEVICE) for bench.py and the
GPU tests.  torch is used for device memory only (allocation, copies, the current stream): the arrays are
handed to libdirect_ddp.so as raw device pointers, exactly as a C++ caller would hand over hipMalloc'ed
buffers (include/direct_ddp.h, direct_ddp_batch_in_t / direct_ddp_batch_out_t)."""
import numpy as np

from . import abi

IN_FIELDS = ("n_seg", "x0", "xd", "T0", "n_planes", "planes", "seeds", "init_bez", "init_poly", "infeas_in")


class DeviceBatch:
    """A HostBatch uploaded once; `cin` is the direct_ddp_batch_in_t that points at it."""

    def __init__(self, host_batch, device):
        import torch
        self.host = host_batch
        self.tens = {}
        for k in IN_FIELDS:
            a = getattr(host_batch, k, None)
            if k == "T0" and getattr(host_batch, "omit_T0", False):
                a = None   # T0 = NULL: the library allocates the durations on the device
            if a is not None:
                self.tens[k] = torch.from_numpy(np.ascontiguousarray(a)).to(device)
        c = abi.BatchIn()
        c.batch, c.n_seg_max, c.p_max, c.mem = host_batch.batch, host_batch.n_seg_max, host_batch.p_max, abi.MEM_DEVICE
        for k, v in self.tens.items():
            setattr(c, k, v.data_ptr())
        self.cin = c


class DeviceResult:
    """Device-resident result arrays (every field of direct_ddp_batch_out_t); `cout` points at them."""

    def __init__(self, batch, n_seg_max, np_dtype, device):
        import torch
        td = torch.float64 if np.dtype(np_dtype) == np.float64 else torch.float32
        z = lambda shape, dt: torch.zeros(shape, dtype=dt, device=device)
        B, N = batch, n_seg_max
        self.t = dict(rtn=z(B, torch.int32), iter_used=z(B, torch.int32), fwd_passes=z(B, torch.int32),
                      infeas_out=z(B, torch.uint8), line_failed_out=z(B, torch.uint8),
                      cost=z(B, td), costq=z(B, td), jerk_cost=z(B, td), terminal_norm2=z(B, td),
                      opterr=z(B, td), mu=z(B, td), bez=z((B, N, 18), td), poly=z((B, N, 18), td), T=z((B, N), td))
        c = abi.BatchOut()
        c.mem = abi.MEM_DEVICE
        for k, v in self.t.items():
            setattr(c, k, v.data_ptr())
        self.cout = c

    def __getitem__(self, k):
        return self.t[k]

    def to_host(self):
        """abi.HostResult-like namespace of numpy arrays (synchronises)."""
        r = abi.HostResult(1, 1)
        for k, v in self.t.items():
            setattr(r, k, v.cpu().numpy())
        return r
