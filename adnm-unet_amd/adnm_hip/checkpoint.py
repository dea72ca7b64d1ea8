"""Checkpoint compatibility with the reference: train.py:169-178 saves `model.state_dict()` as
<model>_best.pth — with a `module.` prefix on every key when train.py:99-102 wrapped the model in nn.DataParallel — and
validate.py:86 / train.py:210 load it back.  The 992 keys and shapes are part of the drop-in boundary."""
import torch

PREFIX = "module."


def strip_prefix(state_dict):
    """nn.DataParallel's `module.` prefix off every key (a no-op for plain checkpoints)."""
    if state_dict and all(k.startswith(PREFIX) for k in state_dict):
        return {k[len(PREFIX):]: v for k, v in state_dict.items()}
    return dict(state_dict)


def load_reference_checkpoint(model, source, map_location="cpu"):
    """Load a reference `*_best.pth` (path or already-loaded state_dict, with or without the DataParallel prefix) into `model`
    after checking that the two key sets and every shape agree.  Works on a model whose parameters FlatTrainer has already re-homed
    into its flat buffer (load_state_dict copies in place, so the flat buffer receives the values).  Returns the number of tensors."""
    sd = torch.load(source, map_location=map_location) if isinstance(source, (str, bytes)) or hasattr(source, "read") else source
    sd = strip_prefix(sd)
    own = model.state_dict()
    missing, unexpected = sorted(set(own) - set(sd)), sorted(set(sd) - set(own))
    if missing or unexpected:
        raise RuntimeError(f"checkpoint does not match the model: {len(missing)} missing (e.g. {missing[:3]}), {len(unexpected)} unexpected (e.g. {unexpected[:3]})")
    bad = [(k, tuple(sd[k].shape), tuple(own[k].shape)) for k in own if tuple(sd[k].shape) != tuple(own[k].shape)]
    if bad:
        raise RuntimeError(f"checkpoint shapes differ from the model's for {len(bad)} tensors, e.g. {bad[:3]}")
    model.load_state_dict(sd, strict=True)
    return len(sd)


def save_reference_checkpoint(model, path, data_parallel_prefix=False):
    """torch.save(model.state_dict()) as train.py:174 does: logical shapes, contiguous CPU tensors (independent of the flat buffer's
    layout), optionally with the `module.` prefix a DataParallel run of the reference would have written."""
    sd = {(PREFIX + k if data_parallel_prefix else k): v.detach().cpu().contiguous() for k, v in model.state_dict().items()}
    torch.save(sd, path)
    return len(sd)
