"""Generate tests/golden/*.npz from the REFERENCE's own modules (run in the build
container only: ``python -m oracle.gen_golden``).  TEST INFRASTRUCTURE ONLY.

The reference holds no tests or fixtures, so parity is pinned by
outputs of the reference itself, produced here by executing its hot-path modules
from /root/reference (oracle/ref_shim.py) on deterministic inputs and on the
storage-free weights of oracle/weights.py.  Only data (inputs, expected outputs,
probe values) is written; no reference source travels.

Randomness the reference draws implicitly is made explicit and recorded:
``torch.randn_like`` (reparameterize, joint_network_v2.py:473) is patched to pop
recorded eps tensors, ``np.random.rand`` (flip axis, denoiser_v2.py:306) to a
recorded value.
"""
import contextlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_shim, weights  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
N_PROBE = 256


def probe(t, key):
    """Small summary of a big tensor: mean/std + values at fixed pseudo-random sites."""
    a = t.detach().cpu().numpy().astype(np.float64).ravel()
    rng = np.random.default_rng(abs(hash(key)) % (2 ** 31) if False else sum(map(ord, key)))
    idx = rng.integers(0, a.size, size=min(N_PROBE, a.size))
    return {key + "/shape": np.asarray(t.shape, dtype=np.int64),
            key + "/mean": np.float64(a.mean()), key + "/std": np.float64(a.std()),
            key + "/absmax": np.float64(np.abs(a).max()),
            key + "/idx": idx.astype(np.int64), key + "/val": a[idx].astype(np.float32)}


def quantised_image(rng, shape):
    """uint8-quantised pixels in [0,1], as the reference's loader yields (utils/loader.py:57-59)."""
    base = rng.normal(0.5, 0.18, size=shape)
    return torch.from_numpy((np.clip(base, 0, 1) * 255).astype(np.uint8).astype(np.float32) / 255.0)


@contextlib.contextmanager
def scripted_randomness(eps_list, flip_p):
    queue = list(eps_list)
    real_randn_like, real_rand = torch.randn_like, np.random.rand

    def fake_randn_like(t, **kw):
        e = queue.pop(0)
        assert e.shape == t.shape, (e.shape, t.shape)
        return e.to(t.dtype)

    torch.randn_like = fake_randn_like
    np.random.rand = lambda *a: flip_p
    try:
        yield
    finally:
        torch.randn_like, np.random.rand = real_randn_like, real_rand
    assert not queue, "unused eps draws"


def load_into(module, sd, prefix):
    own = module.state_dict()
    sub = {k[len(prefix):]: v.clone() for k, v in sd.items() if k.startswith(prefix)}
    assert set(own) == set(sub), (sorted(set(own) ^ set(sub))[:8])
    for k, v in own.items():
        assert tuple(v.shape) == tuple(sub[k].shape), (k, v.shape, sub[k].shape)
    module.load_state_dict(sub, strict=True)


def make_cfg(ref):
    C = ref.params.ConfigValue
    cfg = ref.cfg.base()
    cfg[C.ALGORITHM] = ref.params.NoiseAlgorithm.SELFSUPERVISED_DENOISING
    cfg[C.NOISE_STYLE] = "gaussian"
    cfg[C.NOISE_VALUE] = ref.params.NoiseValue.UNKNOWN_VARIABLE
    cfg[C.IMAGE_CHANNELS] = 1
    cfg[C.NMS] = 18
    cfg[C.BB] = 24
    ref.cfg.infer(cfg, model_only=True)
    return cfg


def make_data(ref, inp, target):
    M = ref.DetectionDataset.Metadata
    b = inp.shape[0]
    meta = {M.GT: [], M.INDEXES: torch.arange(b), M.IMAGE_SHAPE: torch.tensor([list(inp.shape[1:])] * b)}
    hm = torch.zeros_like(inp)
    return [inp, target, hm, hm.clone(), meta]


def main():
    os.makedirs(OUT, exist_ok=True)
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref = ref_shim.load()
    P = ref.params.PipelineOutput
    rng = np.random.default_rng(20261003)
    sd_all = weights.make_state(weights.denoiser_shapes(), seed=0)

    # ---- 0. state-dict layout of the reference Denoiser ------------------------------
    cfg = make_cfg(ref)
    den = ref.Denoiser(cfg, device="cpu", mode="joint")
    layout = {k: (list(v.shape) if hasattr(v, "shape") else "cfg") for k, v in den.state_dict().items()}
    with open(os.path.join(OUT, "state_layout.json"), "w") as f:
        json.dump(layout, f, indent=0, sort_keys=True)
    load_into(den.models["denoiser_model"], sd_all, "denoiser_model.")
    load_into(den.models["sigma_estimation_model"], sd_all, "sigma_estimation_model.")
    jn = den.models["denoiser_model"]
    sg = den.models["sigma_estimation_model"]

    # ---- 1. blind-spot U-Net forward, with per-block probes --------------------------
    x = quantised_image(rng, (2, 1, 64, 64))
    taps = {}
    hooks = []
    db = jn.denoise_branch
    for name in ("encode_block_1", "encode_block_2", "encode_block_3", "encode_block_4",
                 "encode_block_5", "encode_block_6", "decode_block_1"):
        hooks.append(getattr(db, name).register_forward_hook(
            lambda m, i, o, name=name: taps.__setitem__(name, o.detach().clone())))
    with torch.no_grad():
        out_stats, _ = db(x)
    for h in hooks:
        h.remove()
    d = {"x": x.numpy(), "out_stats": out_stats.numpy()}
    for k, v in taps.items():
        d.update(probe(v, k))
    # blind-spot property of the reference: perturb one pixel, that pixel's output must not move
    x2 = x.clone()
    x2[0, 0, 20, 37] += 0.25
    with torch.no_grad():
        out2, _ = db(x2)
    d["blindspot_same"] = np.asarray(bool(torch.equal(out2[0, :, 20, 37], out_stats[0, :, 20, 37])))
    np.savez_compressed(os.path.join(OUT, "unet_fwd.npz"), **d)
    print("unet_fwd ok; blind-spot holds in reference:", d["blindspot_same"])

    # ---- 2. sigma net and detector, stand-alone --------------------------------------
    with torch.no_grad():
        y_sigma = sg(x)
    z = torch.from_numpy(rng.normal(0.4, 0.3, size=(3, 1, 64, 64)).astype(np.float32))
    zf = torch.from_numpy(rng.normal(0.4, 0.3, size=(1, 1, 96, 80)).astype(np.float32))
    det = jn.detector
    det.eval(); det.unfill()
    with torch.no_grad():
        det_eval_unfilled = det(z)
    stride = det.fill()
    with torch.no_grad():
        det_eval_filled = det(zf)
        det_eval_filled64 = det(z[:1])
    det.unfill()
    det.train()
    det_train_unfilled = det(z).detach()
    bn_after = {k: v.clone().numpy() for k, v in det.state_dict().items() if "running" in k or "num_batches" in k}
    load_into(jn, sd_all, "denoiser_model.")  # restore running stats
    np.savez_compressed(os.path.join(OUT, "parts.npz"), x=x.numpy(), sigma_out=y_sigma.numpy(),
                        z=z.numpy(), zf=zf.numpy(), det_eval_unfilled=det_eval_unfilled.numpy(),
                        det_eval_filled=det_eval_filled.numpy(), det_eval_filled64=det_eval_filled64.numpy(),
                        det_train_unfilled=det_train_unfilled.numpy(), fill_stride=np.asarray(stride),
                        **{"bn_after/" + k: v for k, v in bn_after.items()})
    print("parts ok; fill() returned stride", stride,
          "| centre of filled == unfilled:", float((det_eval_filled64[0, 0, 32, 32] - det_eval_unfilled[0, 0, 0, 0]).abs()))

    # ---- 3. joint pipeline, train step (B=4) -----------------------------------------
    B = 4
    inp = quantised_image(rng, (B, 1, 64, 64))
    target = torch.tensor([[1.0], [-1.0], [-1.0], [0.25]])
    eps = torch.from_numpy(rng.normal(size=(B, 1, 64, 64)).astype(np.float32))
    eps_f = torch.from_numpy(rng.normal(size=(B, 1, 64, 64)).astype(np.float32))
    for flip_p, tag in ((0.3, "w"), (0.8, "h")):
        load_into(jn, sd_all, "denoiser_model.")
        load_into(sg, sd_all, "sigma_estimation_model.")
        den.train(); den.unfill(); den.zero_grad()
        with scripted_randomness([eps, eps_f], flip_p):
            o = den.run_pipeline(make_data(ref, inp.clone(), target.clone()), 0.75, 0.01, train=True)
        torch.mean(o[P.LOSS]).backward()
        d = {"inp": inp.numpy(), "target": target.numpy(), "eps": eps.numpy(), "eps_flip": eps_f.numpy(),
             "flip_p": np.asarray(flip_p), "alpha": np.asarray(0.75), "tau": np.asarray(0.01),
             "LOSS": o[P.LOSS].detach().numpy(), "DENOISE_LOSS": o[P.DENOISE_LOSS].detach().numpy(),
             "DETECT_LOSS": o[P.DETECT_LOSS].detach().numpy(), "AUG_LOSS": o[P.AUG_LOSS].detach().numpy(),
             "DETECT": o[P.DETECT].detach().numpy(), "IMG_MU": o[P.IMG_MU].detach().numpy(),
             "IMG_DENOISED": o[P.IMG_DENOISED].detach().numpy(),
             "NOISE_STD_DEV": o[P.NOISE_STD_DEV].detach().numpy(),
             "MODEL_STD_DEV": o[P.MODEL_STD_DEV].detach().numpy()}
        nograd = []
        for name, p in den.models.named_parameters():
            if p.grad is None:
                nograd.append(name)
            else:
                d.update(probe(p.grad, "grad/" + name))
                d["grad/" + name + "/norm"] = np.float64(p.grad.double().norm())
        d["nograd"] = np.asarray(nograd)
        for k, v in jn.detector.state_dict().items():
            if "running" in k or "num_batches" in k:
                d["bn_after/" + k] = v.clone().numpy()
        np.savez_compressed(os.path.join(OUT, "joint_train_%s.npz" % tag), **d)
        print("joint_train", tag, "LOSS", o[P.LOSS].detach().ravel().tolist(), "params without grad:", len(nograd))

    # ---- 4. joint pipeline, eval (filled) on a padded square micrograph --------------
    load_into(jn, sd_all, "denoiser_model.")
    load_into(sg, sd_all, "sigma_estimation_model.")
    S = 128
    inp_e = quantised_image(rng, (1, 1, S, S))
    eps_e = torch.from_numpy(rng.normal(size=(1, 1, S, S)).astype(np.float32))
    den.eval(); den.fill()
    with torch.no_grad(), scripted_randomness([eps_e], 0.0):
        o = den.run_pipeline(make_data(ref, inp_e.clone(), torch.zeros(1, 1)), train=False)
    den.unfill()
    score = o[P.DETECT][0, 0].numpy()
    s_ref, c_ref = ref.non_maximum_suppression(score.copy(), 18, set(), 0.02)
    s_r5, c_r5 = ref.non_maximum_suppression(score.copy(), 5, set(), 0.02)
    np.savez_compressed(os.path.join(OUT, "joint_eval.npz"), inp=inp_e.numpy(), eps=eps_e.numpy(),
                        LOSS=o[P.LOSS].numpy(), DETECT=o[P.DETECT].numpy(), IMG_MU=o[P.IMG_MU].numpy(),
                        IMG_DENOISED=o[P.IMG_DENOISED].numpy(), NOISE_STD_DEV=o[P.NOISE_STD_DEV].numpy(),
                        MODEL_STD_DEV=o[P.MODEL_STD_DEV].numpy(),
                        nms18_scores=s_ref, nms18_coords=c_ref, nms5_scores=s_r5, nms5_coords=c_r5)
    print("joint_eval ok: picks r=18:", len(s_ref), " r=5:", len(s_r5),
          " score range", float(score.min()), float(score.max()))

    # ---- 5. denoise-only (ssdn) pipeline ---------------------------------------------
    den_dn = ref.Denoiser(cfg, device="cpu", mode="denoise")
    load_into(den_dn.models["denoiser_model"], sd_all, "denoiser_model.")
    load_into(den_dn.models["sigma_estimation_model"], sd_all, "sigma_estimation_model.")
    den_dn.eval()
    with torch.no_grad(), scripted_randomness([eps[:2]], 0.0):
        o = den_dn.run_pipeline(make_data(ref, inp[:2].clone(), target[:2].clone()))
    np.savez_compressed(os.path.join(OUT, "ssdn_eval.npz"), inp=inp[:2].numpy(), LOSS=o[P.LOSS].numpy(),
                        IMG_MU=o[P.IMG_MU].numpy(), IMG_DENOISED=o[P.IMG_DENOISED].numpy(),
                        NOISE_STD_DEV=o[P.NOISE_STD_DEV].numpy(), MODEL_STD_DEV=o[P.MODEL_STD_DEV].numpy())
    print("ssdn ok")

    # ---- 6. NMS cases from the reference function ------------------------------------
    cases = {}

    def add(name, arr, r, thr):
        arr = np.ascontiguousarray(arr, dtype=np.float32)
        vals = arr[arr > thr]
        assert len(np.unique(vals)) == len(vals), "fixture must have no ties above threshold: " + name
        s, c = ref.non_maximum_suppression(arr.copy(), r, set(), thr)
        cases[name + "/x"] = arr
        cases[name + "/r"] = np.asarray(r)
        cases[name + "/thr"] = np.asarray(thr, dtype=np.float32)
        cases[name + "/scores"] = s
        cases[name + "/coords"] = c
        print("  nms case %-18s %s r=%d thr=%g -> %d picks" % (name, arr.shape, r, thr, len(s)))

    def uniq(shape):  # random map without ties
        n = int(np.prod(shape))
        return (rng.permutation(n).astype(np.float32).reshape(shape) + 1.0) / (n + 1.0)

    add("rand_r1", uniq((48, 56)), 1, 0.02)
    add("rand_r2", uniq((48, 56)), 2, 0.02)
    add("rand_r18", uniq((96, 128)), 18, 0.02)
    add("rand_r5_thr", uniq((64, 64)), 5, 0.6)
    add("all_below", uniq((32, 32)) * 0.01, 3, 0.02)
    blobs = np.zeros((80, 96), dtype=np.float64)
    yy, xx = np.mgrid[0:80, 0:96]
    k = 0
    for cy, cx in ((0, 0), (0, 95), (79, 0), (79, 95), (40, 95), (40, 0), (0, 50), (79, 50), (30, 40), (33, 47), (41, 94)):
        k += 1
        blobs += (0.5 + 0.04 * k) * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / (2 * 3.0 ** 2))
    blobs += uniq((80, 96)) * 1e-3
    add("borders_r6", blobs, 6, 0.02)
    # x-overflow wrap: a strong peak near the right border suppresses column 0 of the NEXT rows
    wrap = uniq((40, 40)) * 0.1
    wrap[10, 38] = 0.99
    wrap[11, 0] = 0.95   # index (10+0+1)*W+0 -> suppressed only through the wrap
    wrap[14, 0] = 0.94   # (13+1, 0): reached via di=3, dmax(3)=3 -> 38+3>=40 -> suppressed
    wrap[16, 0] = 0.93   # (15+1, 0): di=5 needs 38+dmax(5)=38+0 -> 38 < 40 -> NOT suppressed (r=5)
    add("wrap_r5", wrap, 5, 0.02)
    wrap2 = uniq((24, 24)) * 0.1
    wrap2[0, 23] = 0.99  # top-right corner: y underflow clips to row 0, x overflow -> (1,0)
    wrap2[1, 0] = 0.9
    wrap2[23, 23] = 0.98  # bottom-right: overflow lands past the array
    add("wrap_corner_r3", wrap2, 3, 0.02)
    add("tall_r4", uniq((200, 20)), 4, 0.3)
    np.savez_compressed(os.path.join(OUT, "nms_cases.npz"), **cases)

    # ---- 7. LR schedule and detector width -------------------------------------------
    its = np.arange(0, 80001, 400)
    lr = np.asarray([ref.compute_ramped_lrate(int(i), 80000, 0.7, 0.2, 1e-4) for i in its])
    feats = list(jn.detector.detector.features.features.children())
    np.savez_compressed(os.path.join(OUT, "misc.npz"), lr_iters=its, lr=lr,
                        det_width=np.asarray(ref.insize_from_outsize(feats, 1)))
    print("misc ok; detector width", ref.insize_from_outsize(feats, 1))


if __name__ == "__main__":
    main()
