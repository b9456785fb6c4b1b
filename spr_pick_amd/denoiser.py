"""``Denoiser``: the reference's pipeline object (spr_pick/denoiser_v2.py:36-870) on libsprk.so.

Same constructor, ``models`` / ``_models`` dictionaries, ``run_pipeline`` / ``forward`` / ``fill`` /
``unfill`` / ``state_dict`` / ``from_state_dict`` / ``config_name`` surface and the same output
dictionary, for the configuration BASELINE.json runs: 1 channel, gaussian noise model,
``--noise_value var`` (sigma network), modes "joint" (``_new_pipeline`` :253-589) and "denoise"
(``_ssdn_pipeline`` :598-849).  Differences that are deliberate:

  * the random draws the reference takes implicitly (eps of the reparameterisation, the flip axis)
    can be passed in (``eps=``, ``eps_flip=``, ``flip_p=``) so runs can be replayed exactly;
  * the PU loss never leaves the device: labels, masks and counts are device tensors and the binomial
    table is cached per (B, tau) (reference: ``.item()`` + scipy every step), so a training step has no
    host synchronisation and can be captured in a HIP graph (graph_step.py);
  * everything per-pixel is a fused HIP kernel (ops.py).
"""
import numpy as np
import torch
import torch.nn as nn
from scipy import stats

from . import cfg as cfg_mod
from . import ops
from .datasets import DetectionDataset
from .networks import DualNetworkShallow, JointNetwork
from .params import ConfigValue, NoiseValue, Pipeline, PipelineOutput


def _sigmoid(x):
    """clamp(sigmoid(x), 1e-4, 1-1e-4) (denoiser_v2.py:32-34)."""
    return ops.sigmoid_clamp(x)


class PuLoss(nn.Module):
    """BCE on labelled patches + slack * binomial generalised-expectation penalty on unlabelled
    ones (utils/losses.py:303-349).

    Evaluated entirely on the device from the label vector ``y`` (>= 0: labelled with that target, -1:
    unlabelled), with masks instead of boolean indexing and device-side counts: no value ever travels to the
    host, no shape depends on the data, so the step can be captured in a HIP graph and replayed on any batch.
    The reference's ``binom.logpmf(arange(N+1), N, tau)`` (N = number of unlabelled patches, a different length
    every batch) is row N of a (B+1) x (B+1) table computed once per (B, tau); entries k > N are masked out of
    the softmax, which leaves the same N+1 terms."""

    def __init__(self):
        super().__init__()
        self._tables = {}

    def _log_binom_table(self, B, tau, device):
        key = (B, float(tau), str(device))
        if key not in self._tables:
            k = np.arange(0, B + 1)
            rows = np.zeros((B + 1, B + 1), dtype=np.float64)
            for n in range(B + 1):
                rows[n, :n + 1] = stats.binom.logpmf(k[:n + 1], n, tau)
            self._tables[key] = (torch.from_numpy(rows).float().to(device),
                                 torch.arange(0, B + 1, dtype=torch.float32, device=device))
        return self._tables[key]

    def forward(self, tau, p, y, slack=4.0):
        """One launch on the GPU (ops.pu_loss: loss and d loss / d p together); the mask form below states the same
        maths in torch operators (~40 five-microsecond launches forward and as many backward) and is what the CPU
        tests pin against the oracle."""
        if not p.is_cuda:
            raise RuntimeError("PuLoss runs on the GPU only; mask_form() is the torch statement for CPU checks")
        y = y.reshape(-1).to(device=p.device, dtype=torch.float32)
        table, _ = self._log_binom_table(p.numel(), tau, p.device)
        return ops.pu_loss(p, y, table, slack)

    def mask_form(self, tau, p, y, slack=4.0):
        p = p.reshape(-1)
        y = y.reshape(-1).to(device=p.device, dtype=torch.float32)
        B = p.shape[0]
        m_lab = (y >= 0).float()
        m_unl = (y == -1).float()
        n_lab, n_unl = m_lab.sum(), m_unl.sum()
        yl = y * m_lab                                   # 0 where unlabelled (the term is masked anyway)
        bce = -(yl * torch.log(p) + (1 - yl) * torch.log(1 - p))
        loss = (bce * m_lab).sum() / n_lab.clamp_min(1.0)      # 0 when nothing is labelled, as the reference
        q_mu = (p * m_unl).sum()
        q_var = (p * (1 - p) * m_unl).sum()
        table, counts = self._log_binom_table(B, tau, p.device)
        log_binom = table.index_select(0, n_unl.long().reshape(1))[0]
        valid = counts <= n_unl
        logits = (-0.5 * (q_mu - counts) ** 2 / (q_var + 1e-7)).masked_fill(~valid, float("-inf"))
        q = torch.softmax(logits, dim=0)
        return loss + slack * (-(log_binom * q).sum())


class Denoiser(nn.Module):
    MODEL = "denoiser_model"
    SIGMA_ESTIMATOR = "sigma_estimation_model"
    ESTIMATED_SIGMA = "estimated_sigma"
    PROB_ESTIMATOR = "detector_model"
    MIN_HALO = 352        # receptive field of blind-spot U-Net + filled detector, rounded up to 32 (_tiled_networks)

    def __init__(self, cfg, device=None, mode=None):
        super().__init__()
        self.device = torch.device(device) if device else torch.device("cuda")
        if self.device.type != "cuda":
            raise RuntimeError("spr_pick_amd.Denoiser runs on the GPU only (device=%s); use the oracle/ "
                               "restatement for CPU checks" % self.device)
        if self.device.index is not None and self.device.index != torch.cuda.current_device():
            torch.cuda.set_device(self.device)   # one process drives one GPU: its kernels go to THIS card's streams
        self.cfg = cfg
        self.mode = mode
        self.models = nn.ModuleDict()
        self._models = nn.ModuleDict()
        self.init_networks()
        self.l_params = nn.ParameterDict()
        self.init_l_params()
        self._pu = PuLoss()

    def init_networks(self):
        c = self.cfg
        in_channels = c[ConfigValue.IMAGE_CHANNELS]
        if in_channels != 1:
            raise NotImplementedError("only IMAGE_CHANNELS == 1 (micrographs) is on the hot path")
        self.add_model(Denoiser.MODEL, JointNetwork(in_channels=in_channels, out_channels=2,
                                                    blindspot=c[ConfigValue.BLINDSPOT], detect=True))
        if c[ConfigValue.PIPELINE] == Pipeline.SSDN and c[ConfigValue.NOISE_VALUE] == NoiseValue.UNKNOWN_VARIABLE:
            self.add_model(Denoiser.SIGMA_ESTIMATOR, DualNetworkShallow(in_channels=in_channels, out_channels=1,
                                                                        blindspot=False, detect=False))

    def set_conv_dtype(self, dtype):
        """"f32" (default) | "bf16" | "f16" | "mixed16": operand precision of the U-Nets' MFMA convolutions AND storage type
        of the activation / activation-gradient tensors between their layers (BASELINE configs[4]; networks.set_conv_dtype;
        "<type>/operands": fp32 tensors, operands rounded on the way in).  Parameters, their gradients, the optimiser, the
        detector and everything behind the U-Nets' output convolutions stay fp32."""
        from .networks import set_conv_dtype
        self.conv_dtype = dtype
        return set_conv_dtype(self, dtype)

    def fill(self, stride=1):
        return self.models[Denoiser.MODEL].fill(stride=stride)

    def unfill(self):
        return self.models[Denoiser.MODEL].unfill()

    def init_l_params(self):
        c = self.cfg
        if c[ConfigValue.PIPELINE] == Pipeline.SSDN and c[ConfigValue.NOISE_VALUE] == NoiseValue.UNKNOWN_CONSTANT:
            self.l_params[Denoiser.ESTIMATED_SIGMA] = nn.Parameter(torch.zeros((1, 1, 1, 1), device=self.device))

    def get_model(self, model_id, parallelised=True):
        return (self.models if parallelised else self._models)[model_id]

    def add_model(self, model_id, model, parallelise=False):
        if parallelise:
            raise NotImplementedError("nn.DataParallel is never enabled by the reference (denoiser_v2.py:170); "
                                      "multi-GPU here is one process per GPU (spr_pick_amd.distributed)")
        self._models[model_id] = model
        model.to(self.device)
        self.models[model_id] = model

    def forward(self, data):
        outputs = self.run_pipeline([data])
        return outputs[PipelineOutput.IMG_DENOISED]

    def run_pipeline(self, data, alpha=0, tau=0, train=True, **kwargs):
        if self.cfg[ConfigValue.PIPELINE] == Pipeline.SSDN and self.mode == "denoise":
            return self._ssdn_pipeline(data, **kwargs)
        if self.mode == "joint":
            return self._new_pipeline(data, alpha, tau, train=train, **kwargs)
        raise NotImplementedError("Unsupported processing pipeline")

    # ------------------------------------------------------------------------------------------
    def _noise_std(self, noisy_in):
        c = self.cfg
        nv = c[ConfigValue.NOISE_VALUE]
        if nv == NoiseValue.UNKNOWN_CONSTANT:
            est = self.l_params[Denoiser.ESTIMATED_SIGMA].expand(noisy_in.shape[0], 1, 1, 1)
        elif nv == NoiseValue.UNKNOWN_VARIABLE:
            est = self.models[Denoiser.SIGMA_ESTIMATOR](noisy_in)
            if est.shape[1] == 1 and est[0].numel() <= ops.NOISE_STD_MAX_PIXELS and est.dtype == torch.float32:
                return ops.noise_std_from_map(est)      # patches: mean, shift, softplus and offset in one launch
            est = torch.mean(est, dim=(2, 3), keepdim=True)
        else:
            raise NotImplementedError("NoiseValue.KNOWN is not reachable from the joint pipeline "
                                      "(noise_params_in is undefined in the reference, denoiser_v2.py:406)")
        return torch.nn.functional.softplus(est - 4.0) + 1e-3

    def _labels_on_device(self, target):
        """Patch labels [B,1] -> device.  A host tensor goes through pinned memory, asynchronously: a pageable
        ``.to(device)`` is a synchronous copy, i.e. a full device sync in the middle of every step (measured in
        round 1: the host sat in it for half of the step).  A device tensor (graph-captured step: a static buffer
        the caller refills) is used as is."""
        t = target if torch.is_tensor(target) else torch.as_tensor(target)
        if t.device.type == "cuda":
            return t
        return t.detach().float().pin_memory().to(self.device, non_blocking=True)

    def _tiled_networks(self, inp, eps, tile, halo):
        """Halo-tiled evaluation of the two networks of the filled pipeline: the micrograph is cut
        into interior blocks of ``tile`` pixels, each evaluated inside a square window of tile + 2*halo pixels (shifted
        inwards at the image borders, so a window edge is either >= halo away from the block or IS the image edge),
        and only the block is kept.  Peak memory follows the window, not the micrograph.

        ``halo`` must cover the receptive field: blind-spot U-Net 315 px (upwards, through the five shifted pools,
        the scale-32 bottleneck and the five nearest-neighbour up-samplings: 4+1, +8+2, +16+4, +32+8, +64+16, +64 at
        scale 32, then +16+64, +8+32, +4+16, +2+8, +1+4, +1 for the blind-spot shift; in every direction after the
        four rotations) + 31 px of the filled detector = 346 -> MIN_HALO 352 (window offsets stay multiples of 32, so
        pooling grids and Winograd tiles of a window coincide with the whole image's).  With the kernel choice pinned
        (networks.Conv2d: _lib.DT_PIN under no_grad) a window whose planes have the divisibility of the whole image's
        (sizes that are multiples of 1024) at an offset that is a multiple of 64 (even at scale 32: the 2x2 output
        tiles of the deepest Winograd level coincide) is computed by the same arithmetic in the same order: the tiled
        result is then BIT-IDENTICAL to the whole-image one, so the picks are identical (tests/test_gpu_pipeline.py:
        tile 1024 / halo 512 and tile 1280 / halo 384 on 3072^2); for other sizes and offsets the deepest levels take
        the direct instead of the Winograd kernel, or another tile phase, and results agree to fp32 rounding.
        Everything after the networks (likelihood, posterior mean, sigmoid, NMS) runs on the assembled full-size tensors
        exactly as in the whole-image path, incl. the GLOBAL mean behind noise_std.
        Returns (net_out [1,2,H,W], detector logits [1,1,H,W], noise_std [1,1,1,1])."""
        B, _, H, W = inp.shape
        S = tile + 2 * halo
        if tile % 32 or halo % 32 or H % 32 or W % 32:
            raise ValueError("tiled inference needs tile, halo and the image size to be multiples of 32")
        if halo < self.MIN_HALO:
            raise ValueError("halo %d is smaller than the networks' receptive field (%d)" % (halo, self.MIN_HALO))
        if H < S or W < S:
            raise ValueError("image %dx%d is smaller than one window (%d): evaluate it whole" % (H, W, S))
        model, sigma = self.models[Denoiser.MODEL], self.models[Denoiser.SIGMA_ESTIMATOR]
        if eps is None:
            eps = torch.randn((B, 1, H, W), dtype=inp.dtype, device=inp.device)
        net_out = torch.empty((B, 2, H, W), dtype=torch.float32, device=inp.device)
        logits = torch.empty((B, 1, H, W), dtype=torch.float32, device=inp.device)
        est = torch.empty((B, 1, H, W), dtype=torch.float32, device=inp.device)
        for y0 in range(0, H, tile):
            y1 = min(y0 + tile, H)
            wy = min(max(y0 - halo, 0), H - S)
            for x0 in range(0, W, tile):
                x1 = min(x0 + tile, W)
                wx = min(max(x0 - halo, 0), W - S)
                win = inp[:, :, wy:wy + S, wx:wx + S].contiguous()
                o, d = model(win, eps=eps[:, :, wy:wy + S, wx:wx + S].contiguous())
                e = sigma(win)
                iy, ix = slice(y0 - wy, y1 - wy), slice(x0 - wx, x1 - wx)
                net_out[:, :, y0:y1, x0:x1] = o[:, :, iy, ix]
                logits[:, :, y0:y1, x0:x1] = d[:, :, iy, ix]
                est[:, :, y0:y1, x0:x1] = e[:, :, iy, ix]
                del o, d, e, win
        noise_std = torch.nn.functional.softplus(torch.mean(est, dim=(2, 3), keepdim=True) - 4.0) + 1e-3
        return net_out, logits, noise_std

    def _check_style(self):
        """-> ops.NOISE_* of the configured likelihood (denoiser_v2.py:405-424): gaussian, or the signal-dependent
        poisson approximation with an unknown parameter."""
        style = self.cfg[ConfigValue.NOISE_STYLE]
        if style is not None and style.startswith("gauss"):
            return ops.NOISE_GAUSSIAN
        if style is not None and style.startswith("poisson"):
            return ops.NOISE_POISSON
        raise NotImplementedError("noise style %r: the reference's likelihood covers gauss* and poisson* only "
                                  "(denoiser_v2.py:405-424)" % (style,))

    def _new_pipeline(self, data, alpha, tau, train, eps=None, eps_flip=None, flip_p=None, tile=None, halo=512, **kwargs):
        if not len(data) > 2:
            return None  # as the reference: the joint pipeline needs the full batch list (denoiser_v2.py:261)
        style = self._check_style()
        inp, target = data[DetectionDataset.INPUT], data[DetectionDataset.TARGET]
        hm = data[DetectionDataset.HM]
        metadata = data[DetectionDataset.METADATA]
        gt = metadata.get(DetectionDataset.Metadata.GT) if isinstance(metadata, dict) else None
        inp = inp.to(self.device, dtype=torch.float32)
        if torch.is_tensor(hm):
            hm = hm.to(self.device)
        model = self.models[Denoiser.MODEL]

        if train:
            p = np.random.rand() if flip_p is None else flip_p
            axis = -1 if p <= 0.5 else -2
            (net_out, hm_p), (_, hm_p_f) = model.forward_pair(inp, inp.flip(axis), eps, eps_flip)
            hm_p = _sigmoid(hm_p)
            hm_p_f = _sigmoid(hm_p_f)         # of the flipped batch, not flipped back: ops.joint_loss reads it mirrored
            pred_loss = self._pu(tau, hm_p, self._labels_on_device(target))
        elif tile:
            net_out, hm_p, noise_std = self._tiled_networks(inp, eps, int(tile), int(halo))
            hm_p = _sigmoid(hm_p)
        else:
            net_out, hm_p = model(inp, eps=eps)
            hm_p = _sigmoid(hm_p)

        mu_x = net_out[:, 0:1]
        if not (tile and not train):
            noise_std = self._noise_std(inp)
        loss_out, pme_out, net_std_out, ns_map = ops.ssdn_nll_pme(inp, net_out, noise_std, style)
        if train:
            # consis = mse(hm_p, flip(hm_p_f)); final = alpha * loss_out + (1 - alpha) * pred_loss + 0.1 * consis
            final_loss, consis_loss = ops.joint_loss(loss_out, pred_loss, hm_p, hm_p_f, axis, alpha, 0.1)
        else:
            final_loss, pred_loss, consis_loss = loss_out, 0, 0
        return {
            PipelineOutput.INPUTS: data,
            PipelineOutput.IMG_MU: mu_x,
            PipelineOutput.TARGET: hm,
            PipelineOutput.AUG_LOSS: consis_loss,
            PipelineOutput.LOSS: final_loss,
            PipelineOutput.IMG_DENOISED: pme_out,
            PipelineOutput.DETECT_LOSS: pred_loss,
            PipelineOutput.DENOISE_LOSS: loss_out,
            PipelineOutput.NOISE_STD_DEV: ns_map if style == ops.NOISE_POISSON else noise_std[:, 0],
            PipelineOutput.MODEL_STD_DEV: net_std_out,
            PipelineOutput.DETECT: hm_p,
            PipelineOutput.GT: gt,
        }

    def _ssdn_pipeline(self, data, **kwargs):
        style = self._check_style()
        inp, target = data[DetectionDataset.INPUT], data[DetectionDataset.TARGET]
        inp = inp.to(self.device, dtype=torch.float32)
        if torch.is_tensor(target):
            target = target.to(self.device)
        # the reference calls the whole JointNetwork here too but only uses out_stats
        res = self.models[Denoiser.MODEL].denoise_branch(inp)
        net_out = res[0] if isinstance(res, tuple) else res
        noise_std = self._noise_std(inp)
        loss_out, pme_out, net_std_out, ns_map = ops.ssdn_nll_pme(inp, net_out, noise_std, style)
        return {
            PipelineOutput.INPUTS: data,
            PipelineOutput.IMG_MU: net_out[:, 0:1],
            PipelineOutput.TARGET: target,
            PipelineOutput.IMG_DENOISED: pme_out,
            PipelineOutput.LOSS: loss_out,
            PipelineOutput.NOISE_STD_DEV: ns_map if style == ops.NOISE_POISSON else noise_std[:, 0],
            PipelineOutput.MODEL_STD_DEV: net_std_out,
        }

    # ------------------------------------------------------------------------------------------
    def state_dict(self, params_only=False, **kw):
        sd = super().state_dict(**kw)
        if not params_only:
            sd["cfg"] = self.cfg
        return sd

    @staticmethod
    def from_state_dict(state_dict, mode, device=None):
        den = Denoiser(state_dict["cfg"], device=device, mode=mode)
        den.load_state_dict({k: v for k, v in state_dict.items() if k != "cfg"}, strict=False)
        return den

    def config_name(self):
        return cfg_mod.config_name(self.cfg)
