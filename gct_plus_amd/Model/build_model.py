"""Model factory with the reference's contract (Model/build_model.py:8-14, 42-87):
get_model(args, src_vocab_len, trg_vocab_len, rank) and load_state(model, path, rank), which
accepts checkpoints saved from a bare model or from a DDP wrapper ('module.' prefix)."""
from collections import OrderedDict

import torch

from .cvaetf import Cvaetf
from .vaetf import Vaetf

model_dict = {
    "vaetf": Vaetf,
    "pvaetf": Cvaetf,
    "scavaetf": Cvaetf,
    "pscavaetf": Cvaetf,
}


def extract_params(args, src_vocab_len, trg_vocab_len):
    # note: `variational` is not forwarded by the reference either => always True
    return {
        "src_vocab": src_vocab_len,
        "trg_vocab": trg_vocab_len,
        "N": args.N,
        "d_model": args.d_model,
        "dff": args.d_ff,
        "h": args.H,
        "latent_dim": args.latent_dim,
        "dropout": args.dropout,
        "use_cond2dec": args.use_cond2dec,
        "use_cond2lat": args.use_cond2lat,
        "nconds": len(args.property_list),
        "get_attn": args.get_attn,
    }


def load_checkpoint(path):
    """torch.load with the weights-only unpickler (nothing in the file is executed).  A checkpoint written by the
    reference's trainer holds ONE non-tensor global: `param_groups[0]['lr']` is a numpy float64 scalar
    (Train/trainer1.py:117-127 computes the lr with numpy and writes it into the optimizer), pickled as
    numpy.core.multiarray.scalar (numpy._core... under numpy 2) + numpy.dtype.  Exactly those names are allow-listed;
    anything else in a file is still refused."""
    import numpy as np
    try:
        from numpy._core.multiarray import scalar
    except ImportError:                                  # numpy 1.x
        from numpy.core.multiarray import scalar
    allowed = [scalar, (scalar, "numpy.core.multiarray.scalar"), (scalar, "numpy._core.multiarray.scalar"), np.dtype]
    allowed += [type(np.dtype(t)) for t in (np.float64, np.float32, np.int64, np.int32)]
    with torch.serialization.safe_globals(allowed):
        return torch.load(path, map_location=torch.device("cpu"), weights_only=True)


def load_state(model, model_path, rank=0):
    ckpt = load_checkpoint(model_path)
    state = ckpt["model_state_dict"] if isinstance(ckpt, dict) and "model_state_dict" in ckpt else ckpt
    if next(iter(state.keys())).split(".")[0] == "module":
        state = OrderedDict((k[7:], v) for k, v in state.items())
    model.load_state_dict(state)            # strict, as the reference (build_model.py:75)
    return model


def get_model(args, src_vocab_len, trg_vocab_len, rank=0):
    if args.model_type not in model_dict:
        raise KeyError(f"unknown model_type {args.model_type!r}; expected one of {list(model_dict)}")
    model = model_dict[args.model_type](**extract_params(args, src_vocab_len, trg_vocab_len))
    if hasattr(args, "model_path"):
        model = load_state(model, args.model_path, rank)
    return model
