"""A multi-step TRAJECTORY against the oracle (VERDICT r3 missing #5): ten optimiser steps of the trainer's own loop —
HIP-graph replay, FlatGrads, MultiAdam, the learning-rate ramp the reference executes — against ten steps of the
oracle's loop (oracle.pipeline.joint_pipeline + backward + torch.optim.Adam on the CPU) on the SAME ten batches, the
same reparameterisation noise, the same flip draws and the same ramp.  The golden fixtures pin ONE step per flip axis;
this pins LR ramp x Adam x BatchNorm running statistics over consecutive steps (reference loop: train.py:329-338,
417-442; Adam train.py:128-140)."""
import numpy as np
import pytest
import torch

from test_gpu_pipeline import make_cfg

pytestmark = pytest.mark.gpu

STEPS, B = 10, 4
KEYS = ("LOSS", "DENOISE_LOSS", "DETECT_LOSS", "AUG_LOSS", "DETECT")
ITERATIONS = 44          # images: the ramp-up (70 %, as executed) ends at image 30.8 (step 7), the ramp-down (20 %) starts at 35.2 (step 9)


def test_ten_steps_follow_the_oracle_loop(oracle_state):
    from oracle import pipeline as oracle_pipeline
    from spr_pick_amd import Denoiser, graph_step, synthetic
    from spr_pick_amd.params import PipelineOutput as P
    from spr_pick_amd.utils import compute_ramped_lrate

    mics = [synthetic.micrograph(i, size=512, blobs=60) for i in range(2)]
    batches = synthetic.patch_batches(STEPS, B, mics, seed=11, device="cpu")
    g = torch.Generator().manual_seed(5)
    eps = [(torch.randn(B, 1, 64, 64, generator=g), torch.randn(B, 1, 64, 64, generator=g)) for _ in range(STEPS)]
    flips = [float(v) for v in torch.rand(STEPS, generator=g)]
    assert min(flips) <= 0.5 < max(flips)                      # both flip-axis graphs are replayed
    assert any(float(t.max()) >= 0 for _, t in batches) and any(float(t.min()) < 0 for _, t in batches)

    # the oracle's loop (CPU), twice: in fp64 (the reference trajectory) and in fp32 (the precision the reference itself
    # runs in).  How far those two drift apart is the budget ANY correct fp32 implementation needs: the randomly
    # initialised fixture divides by predicted variances near zero and Adam is scale-free, so rounding-level differences
    # grow once the ramp has opened the learning rate (measured: 1e-6 at steps 0-2, 1e-5 .. 1e-4 at steps 3-4, up to
    # 2e-3 at steps 5-9; scratch/r4/traj_sens.py).
    def oracle_loop(dt):
        sd = {k: (v.clone().to(dt) if v.is_floating_point() else v.clone()) for k, v in oracle_state.items()}
        for k, v in sd.items():
            if v.is_floating_point() and "running" not in k:
                v.requires_grad_(True)
        opt_o = torch.optim.Adam([v for v in sd.values() if v.requires_grad], lr=1e-4, betas=(0.9, 0.99))
        lrs, out = [], []
        for i in range(STEPS):
            # train.py:430-442: the two fractions arrive swapped -> 70 % ramp-up, 20 % ramp-down, base 1e-4
            lr = oracle_pipeline.trainer_lrate(i * B, ITERATIONS)
            lrs.append(lr)
            for grp in opt_o.param_groups:
                grp["lr"] = lr
            opt_o.zero_grad(set_to_none=True)
            inp, tgt = batches[i]
            r = oracle_pipeline.joint_pipeline(sd, inp.to(dt), tgt.to(dt), 0.75, 0.01, True, eps[i][0].to(dt), eps[i][1].to(dt),
                                               flips[i])
            r["LOSS"].mean().backward()
            opt_o.step()
            out.append({k: r[k].detach().double().clone() for k in KEYS})
        return lrs, out, sd

    lrs, ref, sd = oracle_loop(torch.float64)
    _, ref32, sd32 = oracle_loop(torch.float32)
    assert lrs[0] == 0.0 and max(lrs) > 9e-5 and lrs[-1] < max(lrs)      # the ramp rises, peaks and falls inside the run

    def deviation(got, want, k):
        # AUG_LOSS = mean((p - p_flip)^2) is a small difference of probabilities: measured in units of a squared
        # probability (it enters LOSS with weight 0.1); everything else relative to its own magnitude
        scale = 1.0 if k == "AUG_LOSS" else max(float(want.abs().max()), 1e-3)
        return float((got.double().reshape(want.shape) - want).abs().max()) / scale

    drift, env = [], {k: 0.0 for k in KEYS}
    for i in range(STEPS):
        for k in KEYS:
            env[k] = max(env[k], deviation(ref32[i][k], ref[i][k], k))
        drift.append(dict(env))                                # running maximum: the envelope of the fp32 oracle's own drift

    # the trainer's loop (GPU): graph replay + MultiAdam, the learning rate as DenoiserTrainer sets it
    den = Denoiser(make_cfg(), device="cuda:0", mode="joint")
    missing, unexpected = den.load_state_dict({"models." + k: v for k, v in oracle_state.items()}, strict=False)
    assert not unexpected
    den.train(); den.unfill()
    init = {k: v.clone() for k, v in den.state_dict().items() if torch.is_tensor(v)}
    st = graph_step.GraphedTrainStep(den, B, 64, 0.75, 0.01, draw_eps=False, eager_warmup=1)
    opt = graph_step.make_adam([p for p in den.parameters() if p.requires_grad], lr=1e-4, betas=(0.9, 0.99))
    st.prepare(batches[0][0].cuda(), batches[0][1], eps[0][0].cuda(), eps[0][1].cuda())
    assert set(st._graphs) == {"w", "h"} and st.fallback_reason is None
    den.load_state_dict(init, strict=False)          # prepare() advanced the BatchNorm buffers; parameters untouched
    rows, bad = [], []
    for i in range(STEPS):
        lr = compute_ramped_lrate(i * B, ITERATIONS, 0.7, 0.2, 1e-4)
        assert abs(lr - lrs[i]) <= 1e-12 * max(lrs[i], 1e-30) + 1e-18
        graph_step.set_lr(opt, lr)
        inp, tgt = batches[i]
        # the parameters this step's forward pass runs on, for the LOCAL check below
        here = {k[len("models."):]: v.detach().cpu().double() for k, v in den.state_dict().items()
                if torch.is_tensor(v) and k.startswith("models.") and v.is_floating_point()}
        o = st(inp.cuda(), tgt, flip_p=flips[i], eps=eps[i][0].cuda(), eps_flip=eps[i][1].cuda())
        got = {"LOSS": o[P.LOSS], "DENOISE_LOSS": o[P.DENOISE_LOSS], "DETECT_LOSS": o[P.DETECT_LOSS].reshape(()),
               "AUG_LOSS": o[P.AUG_LOSS].reshape(()), "DETECT": o[P.DETECT]}
        got = {k: v.detach().cpu().clone() for k, v in got.items()}
        # LOCAL parity at every step (no chaos in this statement: one forward pass): the fp64 oracle evaluated on the
        # parameters the HIP trajectory has reached gives the losses and the score map the replayed graph gave
        with torch.no_grad():
            sd_here = {k: (here[k] if k in here else v.clone()) for k, v in sd.items()}     # (the oracle updates buffers in place)
            r_here = oracle_pipeline.joint_pipeline(sd_here, inp.double(), tgt.double(), 0.75, 0.01, True, eps[i][0].double(),
                                                    eps[i][1].double(), flips[i])
        local = {k: deviation(got[k], r_here[k].detach().double(), k) for k in KEYS}
        row = {}
        for k in KEYS:
            d = deviation(got[k], ref[i][k], k)
            # TRAJECTORY against the fp64 oracle's own: the first five steps within 1e-3; later, where two correct fp32 runs
            # cannot stay that close, within 8x the fp32 oracle's own drift from the fp64 one up to that step, never beyond
            # 2e-2 (how many multiples of that drift a given arithmetic lands on is a lottery: the score map at step 9 was
            # at 2.3e-3 with one summation order of the backward-weight partial sums and 8.5e-3 with the next; the local
            # check beside it is what does not move)
            budget = 1e-3 if i < 5 else min(max(1e-3, 8.0 * drift[i][k] + 1e-4), 2e-2)
            row[k] = (d, budget, local[k])
            if d > budget:
                bad.append("step %d %s: %.3e > %.3e" % (i, k, d, budget))
            if local[k] > 1e-4:                  # measured <= 1.6e-5 at every step and quantity
                bad.append("step %d %s: %.3e from the fp64 oracle ON THE SAME PARAMETERS" % (i, k, local[k]))
        rows.append(row)
        opt.step()
    for i, row in enumerate(rows):
        print("step %d lr %.2e  " % (i, lrs[i]) + "  ".join("%s %.1e/%.1e (local %.1e)" % (k, *row[k]) for k in KEYS))
    assert not bad, "; ".join(bad)
    # the first half of the run (before the drift sets in) is held to the plain 1e-3
    assert all(rows[i][k][0] <= 1e-3 for i in range(5) for k in KEYS)

    # after ten steps: BatchNorm buffers within 1e-4, parameters within what ten Adam updates of <= 1e-4 allow
    after = den.state_dict()
    n_bn, bn_worst = 0, 0.0
    for k, v in sd.items():
        got = after["models." + k].detach().cpu()
        if "running_" in k:
            scale = max(float(v.abs().max()), 1.0)
            d = float((got.double() - v.double()).abs().max()) / scale
            d32 = float((sd32[k].double() - v.double()).abs().max()) / scale
            bn_worst = max(bn_worst, d)
            # BatchNorm running averages after ten steps (twenty updates): within 1e-4, or 16x the fp32 oracle's own drift
            # from the fp64 one (measured on MI355X: 4.0e-4 and, one summation order later, 7.2e-4 where the oracle's own
            # drift is 8.5e-5, 2.1e-3 where it is 1.2e-3: the buffers average the activations of the drifting trajectory), never
            # beyond 1e-2
            assert d <= min(max(1e-4, 16.0 * d32 + 1e-5), 1e-2), "%s: %.3e (fp32 oracle vs fp64 oracle: %.3e)" % (k, d, d32)
            assert not torch.equal(got, oracle_state[k]), k + " never moved"
            n_bn += 1
        elif "num_batches" in k:
            assert int(got) == int(v) == 2 * STEPS, (k, int(got), int(v))
    assert n_bn >= 16
    print("trajectory: BatchNorm running averages after %d steps: worst relative deviation %.2e" % (STEPS, bn_worst))
    moved = rel_worst = worst32 = 0.0
    n_off = n_off32 = n_el = 0
    for k, v in sd.items():
        if not v.requires_grad:
            continue
        got = after["models." + k].detach().cpu().double()
        d = (got - v.detach().double()).abs()
        d32 = (sd32[k].detach().double() - v.detach().double()).abs()
        step = (v.detach().double() - oracle_state[k].double()).abs()
        moved = max(moved, float(step.max()))
        n_el += d.numel()
        n_off += int((d > 2e-6).sum())
        n_off32 += int((d32 > 2e-6).sum())
        rel_worst = max(rel_worst, float(d.max()))
        worst32 = max(worst32, float(d32.max()))
    print("trajectory: parameters moved by up to %.2e; vs the fp64 oracle: GPU max %.2e (fp32 oracle %.2e), %d of %d beyond 2e-6 "
          "(fp32 oracle: %d)" % (moved, rel_worst, worst32, n_off, n_el, n_off32))
    assert moved > 2e-4
    # Adam is scale-free: an element whose gradient sits at the rounding floor can move by a full update in either loop.
    # The count of parameters beyond 2e-6 (1 % of the distance travelled) of the fp64 trajectory is held to three times the
    # fp32 oracle's own count (or 0.5 %), the worst element to three times the fp32 oracle's own worst (or half the distance
    # travelled).  Measured on MI355X: 233 631 elements, worst 2.9e-4 (first kernels of round 4) / 602 461, worst 4.0e-4 (last) of 5.4e-4
    # travelled; fp32 oracle: 277 561, worst 2.0e-4
    assert n_off <= max(n_el // 200, 3 * n_off32), (n_off, n_off32, n_el)
    assert rel_worst <= max(0.5 * moved, 3.0 * worst32), (rel_worst, worst32, moved)
