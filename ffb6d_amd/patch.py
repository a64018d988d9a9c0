"""Drop-in patcher: swaps the gfx950 operators into an UNMODIFIED reference FFB6D.

    import models.ffb6d, models.RandLA.RandLANet          # the reference's modules
    from ffb6d_amd import patch
    patch.patch_reference(models.ffb6d, models.RandLA.RandLANet)
    model = models.ffb6d.FFB6D(...).cuda().eval()          # forward now runs our kernels

Replaced members (same signatures, same tensor conventions):
    FFB6D.random_sample / FFB6D.nearest_interpolation     ffb6d/models/ffb6d.py:159-194
    Network.random_sample / Network.nearest_interpolation RandLANet.py:86-117
    Building_block.gather_neighbour (static)              RandLANet.py:225-234
    Building_block.relative_pos_encoding (method)         RandLANet.py:216-223
    Att_pooling.forward (method)                          RandLANet.py:243-250

`rows=True` installs the same operators in their channels-last form (ffb6d_amd.ops_cl: float32 or bfloat16 rows, no transposing
copies next to channels_last convolutions, backward without atomics) -- the ones this package's own training step uses; for a
reference model trained with `model.to(memory_format=torch.channels_last)` under torch.autocast(bfloat16).
"""
from . import ops, ops_cl


def _relative_pos_encoding(self, xyz, neigh_idx):
    return ops.relative_pos_encoding(xyz, neigh_idx)


def _att_pooling_forward(self, feature_set):
    att_activation = self.fc(feature_set)
    return self.mlp(ops.att_pool(feature_set, att_activation))


def _att_pooling_forward_rows(self, feature_set):
    att_activation = self.fc(feature_set)
    return self.mlp(ops_cl.att_pool(feature_set, att_activation))


def patch_classes(ffb6d_cls, building_block_cls, att_pooling_cls, network_cls=None, rows=False):
    """Patch the given classes in place; returns a function that restores them."""
    saved = []

    def swap(cls, name, value):
        saved.append((cls, name, cls.__dict__[name]))
        setattr(cls, name, value)

    mod = ops_cl if rows else ops
    swap(ffb6d_cls, "random_sample", staticmethod(mod.random_sample))
    swap(ffb6d_cls, "nearest_interpolation", staticmethod(mod.nearest_interpolation))
    if network_cls is not None:
        swap(network_cls, "random_sample", staticmethod(mod.random_sample))
        swap(network_cls, "nearest_interpolation", staticmethod(mod.nearest_interpolation))
    swap(building_block_cls, "gather_neighbour", staticmethod(ops_cl.gather_neighbour_rows if rows else ops.gather_neighbour))
    swap(building_block_cls, "relative_pos_encoding", _relative_pos_encoding)
    swap(att_pooling_cls, "forward", _att_pooling_forward_rows if rows else _att_pooling_forward)

    def undo():
        for cls, name, value in reversed(saved):
            setattr(cls, name, value)

    return undo


def patch_reference(ffb6d_module, randla_module, rows=False):
    """ffb6d_module = the reference's `models.ffb6d`, randla_module = `models.RandLA.RandLANet`."""
    return patch_classes(ffb6d_module.FFB6D, randla_module.Building_block, randla_module.Att_pooling,
                         getattr(randla_module, "Network", None), rows=rows)
