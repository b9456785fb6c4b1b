"""BASELINE configs[4] in small: the reference's workflow (README.md:79-92: `joint train start`, then `joint eval` of the
final weights) through this package's CLI on a synthetic micrograph set ON DISK, and — what the reference checks by eye
(README.md:88-92) — whether the trained model finds the planted particles: recall / precision of the written
`*_scores.txt` picks against every planted centre (a pick within --bb/2 = 12 px of a still-unmatched centre).
Only 14 of the ~200 particles per micrograph are labelled (README.md:29: part of the particles of a 300x300 sub-region),
the rest is learnt through the positive-unlabelled loss (utils/losses.py:303-349).

Shortened run: 160 000 iterations (images) at batch 16 = 10 000 optimiser steps (~100 s) instead of 80 000 at batch 4 =
20 000 steps; the full-size runs are recorded in profiles/r04_full_pipeline_*.json (full_pipeline.py).

Whether a trained model's picks survive the step from 64x64 patches to whole micrographs is fragile in the reference's
algorithm (the blind-spot U-Net's output level moves with the context its 315 px receptive field sees, and
a BatchNorm on a signal of variance 6e-4 sits behind it); runs are deterministic, so a configuration's outcome changes
only when the arithmetic does — it did in round 4: the 6 000-step configuration pinned until then stopped writing picks
in fp32 when the small-plane convolutions changed their summation order, 8 000 and 10 000 steps give AP 0.92 in fp32 and
mixed16 (scratch/r4/fp_variants.sh).  The patch-level figures below are the robust statement."""
import pytest

pytestmark = pytest.mark.gpu


def test_trained_model_recovers_the_planted_particles(tmp_path):
    import full_pipeline
    out = full_pipeline.main(["--micrographs", "16", "--iterations", "160000", "--batch", "16", "--dtypes", "f32",
                              "--agreement", "f16", "--print-interval", "16000", "--work", str(tmp_path)])
    run = out["runs"]["f32"]
    train, ev = run["train"], run["eval"]
    # the trainer's loop ran the graph-replayed step (no silent eager fallback) and learnt
    assert "HIP-graph replay" in train["step_execution"], train["step_execution"]
    det = [v for _, v in train["detect_loss_curve"]]
    assert det[0] > 50 and det[-1] < 3.0, det                     # PU loss: 90 (p = 0.5 everywhere) -> ~1.3
    assert train["loss_last"]["train/loss"] < train["loss_first"]["train/loss"] - 10
    # the detector itself, in the geometry it was trained in (64x64 patches, unfilled): particles vs background
    pl = run["patch_level_detection"]
    print("patch level:", pl)
    assert pl["auc"] >= 0.98 and pl["recall_at_0.13"] >= 0.9 and pl["false_positive_rate_at_0.13"] <= 0.05, pl
    m = ev["picks_vs_planted_centres"]
    assert m["micrographs"] == 16 and m["n_truth"] == 16 * 200
    print("picks vs planted centres:", m)
    # measured on MI355X: AP 0.925; at the reference exporter's default threshold 0.13 (convert_to_star.py) precision 0.99,
    # recall 0.91.  Floors leave room for other hardware summation orders, not for a model that has not learnt
    assert m["average_precision"] >= 0.85, m
    assert m["best_f1"]["recall"] >= 0.85 and m["best_f1"]["precision"] >= 0.90, m["best_f1"]
    at = m["at"][0.13]
    assert at["recall"] >= 0.80 and at["precision"] >= 0.85, at
    # (labelled centres are ~7 % of the planted ones: recall this high is generalisation, not memorised labels)
    assert "labelled of 3200 planted" in out["workload"]
    # the same fp32-trained checkpoint with fp16 MFMA operands in inference: the TRAINED detector's picks survive
    agree = out["pick_agreement_with_fp32"]
    assert agree["0.13"]["jaccard"] >= 0.95 and agree["0.5"]["jaccard"] >= 0.95, agree
    m16 = out["fp32_checkpoint_evaluated_with_f16_operands"]["picks_vs_planted_centres"]
    assert abs(m16["average_precision"] - m["average_precision"]) <= 0.01, (m16["average_precision"], m["average_precision"])
