"""The evaluation side of the hot path, on the GPU: the reference's validate.py:92-125 runs
model.eval() under no_grad, copies every prediction to numpy and walks SimplifiedEvaluator's Python loops
(datasets/Shanghai_metrics.py:49-152) frame by frame.  Here

  * GraphedForward replays the forward as ONE captured hipGraph with no saved activations (no_grad), and
  * GpuEvaluator keeps the contingency counts and the squared / absolute error sums on the device (one HIP pass per batch,
    csrc/dataio.hip::eval_counts) and only reads a (frames, 18) table back in done().

GpuEvaluator mirrors SimplifiedEvaluator's surface: __init__(seq_len, value_scale, thresholds), evaluate(true_batch, pred_batch),
done() -> {"threshold_metrics": {thr: TP, TN, FP, FN, CSI, POD, HSS}, "FAR", "RMSE", "SSIM", "LPIPS"}, reset().  SSIM is the reference's
cal_ssim (Shanghai_metrics.py:132-152: 11x11 Gaussian windows, valid region, float64) as one HIP pass per batch (csrc/dataio.hip::eval_ssim),
pinned by the reference's own code run with the two cv2 calls replaced by their documented formulas (cv2 is not installed in the build
container).  LPIPS (a downloaded AlexNet) is outside the hot path and not reproduced: its entry is None."""
import ctypes

import numpy as np
import torch

from . import lib, ops


class GraphedForward:
    """model.eval() forward under no_grad, captured once per input shape and replayed (validate.py:101-104)."""

    def __init__(self, model):
        self.model = model
        self._graphs = {}

    @torch.no_grad()
    def __call__(self, x):
        key = (tuple(x.shape), x.dtype, x.device)
        ent = self._graphs.get(key)
        if ent is None:
            was_training = self.model.training
            self.model.eval()
            sx = x.clone()
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    self.model(sx)
            torch.cuda.current_stream().wait_stream(side)
            g = torch.cuda.CUDAGraph()
            # the graph's split GEMM launches get their own uncached workspace (kept with the graph: a training graph replayed on
            # another stream beside this one must not share it); fp8: the quantisation table's rows stay put while the graph lives
            scope = ops.SPLITWS.open_scope(x.device)
            if ops.mfma_precision() == "fp8":
                ops.QUANT.pin(x.device)
            with ops.SPLITWS.capturing(scope):
                with torch.cuda.graph(g):
                    out = self.model(sx)
            self.model.train(was_training)
            ent = self._graphs[key] = (g, sx, out, scope)
        g, sx, out, _ = ent
        sx.copy_(x, non_blocking=True)
        g.replay()
        return out


class GpuEvaluator:
    def __init__(self, seq_len, value_scale, thresholds=(20, 30, 35, 40)):
        self.seq_len, self.value_scale, self.thresholds = seq_len, float(value_scale), [float(t) for t in thresholds]
        if not 1 <= len(self.thresholds) <= 8:
            raise ValueError("1..8 thresholds")
        self._thr = (ctypes.c_float * len(self.thresholds))(*self.thresholds)
        self.reset()

    def reset(self):
        self._tables = []   # one (B, T, 4*nthr+2) device tensor per evaluate() call
        self._ssim = []     # one (B, T) device tensor of SSIM-map sums per evaluate() call (None for frames too small for the window)
        self.total = 0

    def evaluate(self, true_batch, pred_batch):
        """true_batch / pred_batch: (B, T, H, W) or (B, T, 1, H, W) fp32 GPU tensors in [0, 1] (clipped here as the reference does)."""
        t, p = true_batch, pred_batch
        if not (torch.is_tensor(t) and t.is_cuda and torch.is_tensor(p) and p.is_cuda):
            raise RuntimeError("GpuEvaluator runs on GPU tensors only (there is no numpy path here)")
        if t.dim() == 5:
            t, p = t.squeeze(2), p.squeeze(2)
        t, p = t.float().contiguous(), p.float().contiguous()
        B, T, H, W = t.shape
        nthr = len(self.thresholds)
        out = torch.empty((B, T, 4 * nthr + 2), dtype=torch.float32, device=t.device)
        nb = lib.query("adnm_eval_counts_ws_bytes", B * T, H * W, nthr)
        ws = torch.empty(max(int(nb), 16), dtype=torch.uint8, device=t.device)
        lib.call("adnm_eval_counts", t.data_ptr(), p.data_ptr(), out.data_ptr(), self._thr, nthr, self.value_scale, ws.data_ptr(), nb, B * T, H * W,
                 torch.cuda.current_stream().cuda_stream)
        ssim = None
        if H > 10 and W > 10:   # 11 x 11 windows need a valid region
            ssim = torch.empty((B, T), dtype=torch.float32, device=t.device)
            nb2 = lib.query("adnm_eval_ssim_ws_bytes", B * T, H, W)
            ws2 = torch.empty(max(int(nb2), 16), dtype=torch.uint8, device=t.device)
            lib.call("adnm_eval_ssim", t.data_ptr(), p.data_ptr(), ssim.data_ptr(), self.value_scale, ws2.data_ptr(), nb2, B * T, H, W,
                     torch.cuda.current_stream().cuda_stream)
        self._tables.append((out, H * W))
        self._ssim.append((ssim, (H - 10) * (W - 10)))
        self.total += B

    def done(self):
        """Same aggregation as SimplifiedEvaluator.done (Shanghai_metrics.py:218-290)."""
        nthr = len(self.thresholds)
        tabs = [(t.double().cpu().numpy(), hw) for t, hw in self._tables]   # the one device -> host read
        counts = np.concatenate([t[..., :4 * nthr] for t, _ in tabs], axis=0)   # (samples, T, 4*nthr)
        mse = np.concatenate([t[..., 4 * nthr + 1] / hw for t, hw in tabs], axis=0)   # (samples, T)
        mae = np.concatenate([t[..., 4 * nthr] / hw for t, hw in tabs], axis=0)
        metrics, all_far = {}, []
        with np.errstate(divide="ignore", invalid="ignore"):
            for k, thr in enumerate(self.thresholds):
                TP, FN, FP, TN = (counts[..., 4 * k + i].sum() for i in range(4))
                csi, pod = TP / (TP + FP + FN), TP / (TP + FN)
                hss = (2 * (TP * TN - FP * FN)) / (FP ** 2 + FN ** 2 + 2 * TP * TN + (FP + FN) * (TP + TN))
                all_far.append(FP / (TP + FP))
                key = int(thr) if float(thr).is_integer() else thr
                metrics[key] = {"TP": TP, "TN": TN, "FP": FP, "FN": FN, "CSI": csi, "POD": pod, "HSS": hss}
            rmse = float(np.mean(np.sqrt(np.mean(mse, axis=0))))
        return {"threshold_metrics": metrics, "FAR": float(np.mean(all_far)), "RMSE": rmse, "MAE": float(mae.mean()), "MSE": float(mse.mean()),
                "SSIM": (float(np.mean(np.concatenate([s.double().cpu().numpy() / area for s, area in self._ssim], axis=0)))
                         if self._ssim and all(s is not None for s, _ in self._ssim) else None),
                "LPIPS": None}
