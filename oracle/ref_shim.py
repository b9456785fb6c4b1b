"""Loader for the reference's hot-path modules (TEST INFRASTRUCTURE ONLY).

Used by ``oracle/gen_golden.py`` (fixture generation) and by the CPU tests that
cross-check the restatement in ``oracle/`` against the reference when
``/root/reference`` is present (it never is on the GPU box).

The reference package cannot be imported as a whole (``spr_pick/__init__.py``
pulls in cv2 / torchvision / colorlog, which are absent here — ordinary
ModuleNotFoundErrors).  The hot-path modules themselves only
need torch / numpy / scipy, so we register bare package objects (their
``__init__`` files never execute), stub the two absent third-party names that
are imported at module scope but only dereferenced in non-hot-path functions,
and import the modules one by one.  Nothing is copied: the reference files are
executed from where they lie.
"""
import importlib
import os
import sys
import types

REF_ROOT = os.environ.get("SPRK_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "spr_pick"))


_loaded = None


def load():
    """Return a namespace with the reference's hot-path symbols."""
    global _loaded
    if _loaded is not None:
        return _loaded
    if not available():
        raise RuntimeError("reference tree not present at %s" % REF_ROOT)
    sys.dont_write_bytecode = True  # the reference tree is read-only

    def bare(name, path):
        m = types.ModuleType(name)
        m.__path__ = [path]
        sys.modules[name] = m
        return m

    root = bare("spr_pick", REF_ROOT + "/spr_pick")
    for sub in ("utils", "models", "datasets"):
        setattr(root, sub, bare("spr_pick." + sub, "%s/spr_pick/%s" % (REF_ROOT, sub)))
    for stub in ("cv2", "torchvision", "torchvision.utils"):
        if stub not in sys.modules:
            sys.modules[stub] = types.ModuleType(stub)

    imp = importlib.import_module
    for n in ("utils.utils", "utils.data_format", "utils.data", "utils.algorithms",
              "utils.losses", "params", "cfg"):
        imp("spr_pick." + n)
    root.cfg = sys.modules["spr_pick.cfg"]
    root.params = sys.modules["spr_pick.params"]
    root.utils.rotate = sys.modules["spr_pick.utils.data"].rotate
    for n in ("datasets.image_wrapper", "datasets.noise_wrapper"):
        imp("spr_pick." + n)
    ds = sys.modules["spr_pick.datasets"]
    ds.DetectionDataset = sys.modules["spr_pick.datasets.image_wrapper"].DetectionDataset
    ds.NoisyDataset = sys.modules["spr_pick.datasets.noise_wrapper"].NoisyDataset
    for n in ("utility", "feature_extractor", "classifier", "noise_network",
              "noise_estimation_network", "joint_network_v2", "joint_network_v2_shallow",
              "joint_network_v2_shallower"):
        imp("spr_pick.models." + n)
    md = sys.modules["spr_pick.models"]
    fe = sys.modules["spr_pick.models.feature_extractor"]
    md.NoiseNetwork = sys.modules["spr_pick.models.noise_network"].NoiseNetwork
    md.NoiseEstNetwork = sys.modules["spr_pick.models.noise_estimation_network"].NoiseEstNetwork
    md.ResNet6, md.ResNet8, md.ResNet16 = fe.ResNet6, fe.ResNet8, fe.ResNet16
    md.LinearClassifier = sys.modules["spr_pick.models.classifier"].LinearClassifier
    jn = sys.modules["spr_pick.models.joint_network_v2"]
    md.DualNetwork, md.JointNetwork = jn.DualNetwork, jn.JointNetwork
    md.DualNetworkShallow = sys.modules["spr_pick.models.joint_network_v2_shallow"].DualNetworkShallow
    md.DualNetworkShallower = sys.modules["spr_pick.models.joint_network_v2_shallower"].DualNetworkShallower
    imp("spr_pick.denoiser_v2")

    ns = types.SimpleNamespace()
    ns.Denoiser = sys.modules["spr_pick.denoiser_v2"].Denoiser
    ns.denoiser_v2 = sys.modules["spr_pick.denoiser_v2"]
    ns.JointNetwork = jn.JointNetwork
    ns.DualNetwork = jn.DualNetwork
    ns.Detector = jn.Detector
    ns.ShiftConv2d = jn.ShiftConv2d
    ns.DualNetworkShallow = md.DualNetworkShallow
    ns.params = sys.modules["spr_pick.params"]
    ns.cfg = sys.modules["spr_pick.cfg"]
    ns.DetectionDataset = ds.DetectionDataset
    ns.non_maximum_suppression = sys.modules["spr_pick.utils.algorithms"].non_maximum_suppression
    ns.compute_ramped_lrate = sys.modules["spr_pick.utils.utils"].compute_ramped_lrate
    ns.insize_from_outsize = sys.modules["spr_pick.utils.utils"].insize_from_outsize
    ns.rotate = root.utils.rotate
    ns.pu_loss = sys.modules["spr_pick.utils.losses"].pu_loss
    ns.PuLoss = sys.modules["spr_pick.utils.losses"].PuLoss
    _loaded = ns
    return ns
