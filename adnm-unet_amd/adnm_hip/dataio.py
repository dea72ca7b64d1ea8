"""The input side of the hot path: datasets/Shanghai.py:52-59,121 turns every uint8 radar sample
(25, H0, W0) into float32, divides by 255, resizes to (S, S) with torchvision's tensor Resize (bilinear, align_corners=False, no
antialias) on the CPU, and train.py:133-134 copies the float batch to the GPU from pageable memory, synchronously.

RadarIngest moves the BYTES instead (4x less PCIe traffic than floats, and before the resize: the source frames are larger than
the model's input): two pinned host staging buffers and two device byte buffers alternate, the copy of batch i+1 runs on its own
stream beside the compute of batch i, and one HIP kernel (csrc/dataio.hip::radar_ingest) does / 255 + resize + the
(B, T, 1, S, S) layout on the device."""
import numpy as np
import torch

from . import lib


class RadarIngest:
    def __init__(self, batch, frames, H0, W0, size, device, in_frames=5):
        self.shape = (batch, frames, H0, W0)
        self.size, self.in_frames, self.device = size, in_frames, torch.device(device)
        self.pinned = [torch.empty(self.shape, dtype=torch.uint8).pin_memory() for _ in range(2)]
        self.dev = [torch.empty(self.shape, dtype=torch.uint8, device=self.device) for _ in range(2)]
        self.copy_stream = torch.cuda.Stream(device=self.device)
        self.ready = [torch.cuda.Event() for _ in range(2)]
        self.consumed = [torch.cuda.Event() for _ in range(2)]
        self._slot = 0
        self._pending = []            # FIFO of submitted, not yet taken slots (prefetch depth <= 2 = the number of slots)

    def submit(self, batch_u8):
        """Start moving a (B, T, H0, W0) uint8 batch (numpy array or CPU tensor) to the device; returns at once.
        At most two batches may be in flight (one per slot): a third submit() without a take() raises instead of overwriting a
        staging buffer whose host-to-device copy may still be running."""
        if len(self._pending) >= 2:
            raise RuntimeError("RadarIngest.submit(): both staging slots hold batches that were never taken (prefetch depth is 2)")
        s = self._slot
        self._slot ^= 1
        src = torch.from_numpy(batch_u8) if isinstance(batch_u8, np.ndarray) else batch_u8
        if tuple(src.shape) != self.shape or src.dtype != torch.uint8:
            raise RuntimeError(f"RadarIngest: expected uint8 {self.shape}, got {src.dtype} {tuple(src.shape)}")
        self.consumed[s].synchronize()          # the kernel that read this slot's previous contents has finished
        self.pinned[s].copy_(src)               # host memcpy into pinned memory
        with torch.cuda.stream(self.copy_stream):
            self.dev[s].copy_(self.pinned[s], non_blocking=True)
            self.ready[s].record(self.copy_stream)
        self._pending.append(s)

    def take(self):
        """-> (imgs (B, T_in, 1, S, S), targets (B, T_out, 1, S, S)) fp32 on the device, as train.py:133 splits them."""
        if not self._pending:
            raise RuntimeError("RadarIngest.take() without a submitted batch")
        s = self._pending.pop(0)
        B, T, H0, W0 = self.shape
        cur = torch.cuda.current_stream(self.device)
        cur.wait_event(self.ready[s])
        out = torch.empty((B, T, 1, self.size, self.size), dtype=torch.float32, device=self.device)
        lib.call("adnm_radar_ingest", self.dev[s].data_ptr(), out.data_ptr(), B * T, H0, W0, self.size, 1.0 / 255.0, cur.cuda_stream)
        self.consumed[s].record(cur)
        return out[:, :self.in_frames], out[:, self.in_frames:]


def ingest(batch_u8_device, size):
    """One-shot form on bytes already resident on the device: (B, T, H0, W0) uint8 -> (B, T, 1, S, S) fp32."""
    B, T, H0, W0 = batch_u8_device.shape
    if not batch_u8_device.is_cuda or batch_u8_device.dtype != torch.uint8:
        raise RuntimeError("ingest: needs a uint8 GPU tensor")
    src = batch_u8_device.contiguous()
    out = torch.empty((B, T, 1, size, size), dtype=torch.float32, device=src.device)
    lib.call("adnm_radar_ingest", src.data_ptr(), out.data_ptr(), B * T, H0, W0, size, 1.0 / 255.0, torch.cuda.current_stream().cuda_stream)
    return out
