"""Golden vectors of the POISSON likelihood branch (denoiser_v2.py:412-424: signal-dependent noise
sigma_n^2 = max(mu, 1e-3) * estimate) from the REFERENCE's own modules: one joint train step and one filled eval
pass with ``--noise_style poisson``.  Run in the build container only (``python -m oracle.gen_golden_poisson``);
writes tests/golden/joint_poisson.npz.  TEST INFRASTRUCTURE ONLY.  Own random stream: the fixtures of
oracle/gen_golden.py stay bit-identical."""
import os

import numpy as np
import torch

from oracle import ref_shim, weights
from oracle.gen_golden import OUT, load_into, make_cfg, make_data, probe, quantised_image, scripted_randomness


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    ref = ref_shim.load()
    P = ref.params.PipelineOutput
    rng = np.random.default_rng(20261005)
    sd_all = weights.make_state(weights.denoiser_shapes(), seed=0)
    # the randomly initialised U-Net predicts mu ~ 0 +- 0.3: 96 % of the pixels would sit on the floor side of
    # max(mu, 1e-3).  A bias on the mu channel puts about half of them above it, so both sides of the maximum (and the
    # gradient through mu on the upper side) are pinned.  Recorded in the fixture; the tests apply the same bias.
    MU_BIAS = 0.3
    sd_all["denoiser_model.denoise_branch.output_conv.bias"][0] += MU_BIAS
    cfg = make_cfg(ref)
    cfg[ref.params.ConfigValue.NOISE_STYLE] = "poisson"
    den = ref.Denoiser(cfg, device="cpu", mode="joint")
    jn, sg = den.models["denoiser_model"], den.models["sigma_estimation_model"]
    load_into(jn, sd_all, "denoiser_model.")
    load_into(sg, sd_all, "sigma_estimation_model.")
    B = 3
    inp = quantised_image(rng, (B, 1, 64, 64))
    target = torch.tensor([[0.5], [-1.0], [-1.0]])
    eps = torch.from_numpy(rng.normal(size=(B, 1, 64, 64)).astype(np.float32))
    eps_f = torch.from_numpy(rng.normal(size=(B, 1, 64, 64)).astype(np.float32))
    flip_p = 0.7
    den.train(); den.unfill(); den.zero_grad()
    with scripted_randomness([eps, eps_f], flip_p):
        o = den.run_pipeline(make_data(ref, inp.clone(), target.clone()), 0.75, 0.01, train=True)
    torch.mean(o[P.LOSS]).backward()
    d = {"inp": inp.numpy(), "target": target.numpy(), "eps": eps.numpy(), "eps_flip": eps_f.numpy(),
         "flip_p": np.asarray(flip_p), "alpha": np.asarray(0.75), "tau": np.asarray(0.01), "mu_bias": np.asarray(MU_BIAS)}
    for k in ("LOSS", "DENOISE_LOSS", "DETECT_LOSS", "AUG_LOSS", "DETECT", "IMG_MU", "IMG_DENOISED", "NOISE_STD_DEV",
              "MODEL_STD_DEV"):
        d[k] = o[getattr(P, k)].detach().numpy()
    for name, p in den.models.named_parameters():
        if p.grad is not None:
            d.update(probe(p.grad, "grad/" + name))
            d["grad/" + name + "/norm"] = np.float64(p.grad.double().norm())
    mu = o[P.IMG_MU].detach()
    d["frac_mu_below_floor"] = np.float64((mu < 1e-3).double().mean())
    # eval (filled), fresh weights
    load_into(jn, sd_all, "denoiser_model.")
    load_into(sg, sd_all, "sigma_estimation_model.")
    S = 96
    inp_e = quantised_image(rng, (1, 1, S, S))
    eps_e = torch.from_numpy(rng.normal(size=(1, 1, S, S)).astype(np.float32))
    den.eval(); den.fill()
    with torch.no_grad(), scripted_randomness([eps_e], 0.0):
        oe = den.run_pipeline(make_data(ref, inp_e.clone(), torch.zeros(1, 1)), train=False)
    den.unfill()
    d.update({"eval/inp": inp_e.numpy(), "eval/eps": eps_e.numpy()})
    for k in ("LOSS", "DETECT", "IMG_MU", "IMG_DENOISED", "NOISE_STD_DEV", "MODEL_STD_DEV"):
        d["eval/" + k] = oe[getattr(P, k)].numpy()
    np.savez_compressed(os.path.join(OUT, "joint_poisson.npz"), **d)
    print("joint_poisson ok: LOSS", d["LOSS"].ravel().tolist(), "NOISE_STD_DEV shape", d["NOISE_STD_DEV"].shape,
          "mu below 1e-3:", float(d["frac_mu_below_floor"]), "eval NOISE_STD_DEV shape", d["eval/NOISE_STD_DEV"].shape)


if __name__ == "__main__":
    main()
