"""model_type -> forward function (reference Model/forward_propagation1.py:4-48): builds the
masks from the batch (on the batch's device, no host round trip) and calls model.forward by
keyword exactly like the reference."""
from .modules import get_src_mask, get_trg_mask


def _plain(model, batch, pad_id, use_cond2dec):
    trg_in = batch["trg"][:, :-1]
    return model.forward(src=batch["src"], trg=trg_in,
                         src_mask=get_src_mask(batch["src"], pad_id),
                         trg_mask=get_trg_mask(trg_in, pad_id, use_cond2dec))


def _conditioned(model, batch, pad_id, use_cond2dec):
    trg_in = batch["trg"][:, :-1]
    return model.forward(src=batch["src"], trg=trg_in,
                         src_mask=get_src_mask(batch["src"], pad_id, batch["econds"]),
                         trg_mask=get_trg_mask(trg_in, pad_id, use_cond2dec, batch["dconds"]),
                         econds=batch["econds"], dconds=batch["dconds"])


forward_propagation = {
    "vaetf": _plain,
    "scavaetf": _plain,
    "pvaetf": _conditioned,
    "pscavaetf": _conditioned,
}
