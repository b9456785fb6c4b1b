"""Reading and writing ``.training`` / ``.wt`` files (train.py:586-616, :904-945 of the reference).

Layout: ``.training`` = {"denoiser": Denoiser.state_dict() (+ "cfg"), "state": {StateValue: ...},
"optimizer": Adam.state_dict(), "rng": torch CPU RNG state}; ``.wt`` = Denoiser.state_dict() alone.
Both pickle enum members and small helper objects by module path.  Files written by the reference
name them ``spr_pick.params`` / ``spr_pick.utils.utils``; the unpickler below resolves those to
this package's identically named classes, so the reference's checkpoints load here unchanged."""
import pickle

import torch

_RENAMES = {
    "spr_pick.params": "spr_pick_amd.params",
    "spr_pick.utils.utils": "spr_pick_amd.utils",
    "spr_pick.utils": "spr_pick_amd.utils",
}


class _Unpickler(pickle.Unpickler):
    def find_class(self, module, name):
        return super().find_class(_RENAMES.get(module, module), name)


class _PickleModule:
    """The slice of the ``pickle`` module interface torch.load / torch.save use."""
    __name__ = "spr_pick_amd.checkpoint"
    Unpickler = _Unpickler
    Pickler = pickle.Pickler
    HIGHEST_PROTOCOL = pickle.HIGHEST_PROTOCOL
    dump = staticmethod(pickle.dump)
    dumps = staticmethod(pickle.dumps)

    @staticmethod
    def load(f, **kw):
        return _Unpickler(f, **kw).load()

    @staticmethod
    def loads(b, **kw):
        import io
        return _Unpickler(io.BytesIO(b), **kw).load()


def load(path, map_location="cpu"):
    return torch.load(path, map_location=map_location, weights_only=False, pickle_module=_PickleModule)


def save(obj, path):
    torch.save(obj, path)
