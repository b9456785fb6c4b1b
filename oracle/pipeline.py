"""CPU restatement of the reference's Denoiser pipelines, PU loss and LR ramp.

TEST INFRASTRUCTURE ONLY (see oracle/networks.py header).

Reference behaviour restated (file:line under /root/reference/spr_pick):
  _sigmoid                    denoiser_v2.py:32-34
  Denoiser._new_pipeline      denoiser_v2.py:253-589   (mode="joint")
  Denoiser._ssdn_pipeline     denoiser_v2.py:598-849   (mode="denoise")
  pu_loss / PuLoss            utils/losses.py:303-349  (slack=4.0 via PuLoss.forward)
  compute_ramped_lrate        utils/utils.py:50-69; as called at train.py:434-441
The 1-channel, ``--noise_value var`` branch is restated, with the gaussian likelihood (the one
BASELINE.json's configs run) and the poisson one (denoiser_v2.py:412-424).
"""
import numpy as np
import torch
import torch.nn.functional as F
from scipy import stats

from . import networks

MODEL = "denoiser_model."
SIGMA = "sigma_estimation_model."


def sigmoid_clamp(x):
    return torch.clamp(torch.sigmoid(x), min=1e-4, max=1 - 1e-4)


def pu_loss(tau, p, y, slack=4.0):
    """BCE on labelled entries + slack * binomial GE penalty on unlabelled ones."""
    p = p.reshape(-1)
    y = y.reshape(-1)
    lab = y >= 0
    if int(lab.sum()) > 0:
        cls = F.binary_cross_entropy(p[lab], y[lab])
    else:
        cls = 0
    unl = y == -1
    n = int(unl.sum())
    ph = p[unl]
    q_mu = ph.sum()
    q_var = torch.sum(ph * (1 - ph))
    counts = torch.arange(0, n + 1, dtype=torch.float32, device=p.device)
    q = F.softmax(-0.5 * (q_mu - counts) ** 2 / (q_var + 1e-7), dim=0)
    log_binom = torch.from_numpy(stats.binom.logpmf(np.arange(0, n + 1), n, tau)).float().to(p.device)
    ge = -torch.sum(log_binom * q)
    return cls + slack * ge


def noise_std_from_sigma_net(sd, x):
    y = networks.sigma_unet(sd, SIGMA, x)
    y = torch.mean(y, dim=(2, 3), keepdim=True)
    return F.softplus(y - 4.0) + 1e-3


def ssdn_terms(x, mu, a, noise_std, noise_style="gaussian"):
    """noise_std: the remapped estimate softplus(.-4)+1e-3, [B,1,1,1].  gaussian: that IS the noise std;
    poisson (denoiser_v2.py:412-424, unknown parameter): std = sqrt(max(mu, 1e-3) * estimate), per pixel."""
    if noise_style.startswith("poisson"):
        noise_std = (torch.maximum(mu, torch.tensor(1e-3, dtype=mu.dtype)) * noise_std) ** 0.5
    sigma_x = a ** 2
    sigma_n = noise_std ** 2
    sigma_y = sigma_x + sigma_n
    nll = (x - mu) ** 2 / sigma_y + torch.log(sigma_y) - 0.05 * noise_std
    pme = (x * sigma_x + mu * sigma_n) / (sigma_x + sigma_n)
    return nll, pme, sigma_x, noise_std


def joint_pipeline(sd, inp, target, alpha, tau, train, eps, eps_flip=None, flip_p=None,
                   filled=None, taps=None, noise_style="gaussian"):
    """Denoiser._new_pipeline.  ``sd`` has keys with prefixes MODEL / SIGMA.

    eps / eps_flip: the N(0,1) draws of the two JointNetwork passes;
    flip_p: the host uniform draw choosing the flip axis (<=0.5 -> W, else H).
    """
    if filled is None:
        filled = not train
    out, logit = networks.joint_forward(sd, MODEL, inp, eps, filled, train, taps)
    p = sigmoid_clamp(logit)
    res = {}
    if train:
        axis = -1 if flip_p <= 0.5 else -2
        _, logit_f = networks.joint_forward(sd, MODEL, inp.flip(axis), eps_flip, filled, train)
        p_f = sigmoid_clamp(logit_f.flip(axis))
        pred = pu_loss(tau, p, target)
    mu, a = out[:, 0:1], out[:, 1:2]
    noise_std = noise_std_from_sigma_net(sd, inp)
    nll, pme, sigma_x, noise_std = ssdn_terms(inp, mu, a, noise_std, noise_style)
    nll = nll.reshape(nll.shape[0], -1).mean(1, keepdim=True)
    if train:
        consis = F.mse_loss(p, p_f)
        loss = alpha * nll + (1 - alpha) * pred + 0.1 * consis
        res.update(DETECT_LOSS=pred, AUG_LOSS=consis)
    else:
        loss = nll
    res.update(LOSS=loss, DENOISE_LOSS=nll, IMG_MU=mu, IMG_DENOISED=pme, DETECT=p,
               NOISE_STD_DEV=noise_std[:, 0], MODEL_STD_DEV=(sigma_x ** 0.5)[:, 0].unsqueeze(0))
    return res


def ssdn_pipeline(sd, inp):
    """Denoiser._ssdn_pipeline (denoise-only): JointNetwork is still called (eps is
    drawn and the detector runs) but only out_stats is used."""
    out = networks.unet_blindspot(sd, MODEL + "denoise_branch.", inp)
    mu, a = out[:, 0:1], out[:, 1:2]
    noise_std = noise_std_from_sigma_net(sd, inp)
    nll, pme, sigma_x, noise_std = ssdn_terms(inp, mu, a, noise_std)
    nll = nll.reshape(nll.shape[0], -1).mean(1, keepdim=True)
    return dict(LOSS=nll, IMG_MU=mu, IMG_DENOISED=pme, NOISE_STD_DEV=noise_std[:, 0],
                MODEL_STD_DEV=(sigma_x ** 0.5)[:, 0].unsqueeze(0))


def ramped_lrate(i, iteration_count, ramp_up_fraction, ramp_down_fraction, lr):
    """Cosine ramp-up then squared-cosine ramp-down."""
    if ramp_up_fraction > 0.0 and i <= iteration_count * ramp_up_fraction:
        t = (i / ramp_up_fraction) / iteration_count
        lr = lr * (0.5 - np.cos(t * np.pi) / 2)
    if ramp_down_fraction > 0.0:
        start = iteration_count * (1 - ramp_down_fraction)
        if i >= start:
            t = ((i - start) / ramp_down_fraction) / iteration_count
            lr = lr * (0.5 + np.cos(t * np.pi) / 2) ** 2
    return lr


def trainer_lrate(i, iterations, cfg_rampdown=0.7, cfg_rampup=0.2):
    """LR as the reference trainer actually evaluates it (train.py:434-441): base
    1e-4 hard-coded and the two fractions passed in swapped positions."""
    return ramped_lrate(i, iterations, cfg_rampdown, cfg_rampup, 1e-4)
