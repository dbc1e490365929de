"""model_type -> forward function (reference Model/forward_propagation1.py:4-48): builds the
masks from the batch (on the batch's device, no host round trip; the decoder's self-attention mask straight from
the token ids) and calls model.forward by keyword exactly like the reference.

skip_ignored (an extension, default False = the reference's behaviour): tell the model which decoder rows the
reference's loss looks at -- `ys = trg[:, 1:] != pad` (Train/trainer1.py:21-22,97: cross-entropy with
ignore_index = pad) -- so that it does not compute the others (56 % of the decoder rows at MOSES-like lengths).  Loss
and every gradient are unchanged; the logits of the ignored rows are not the reference's (nothing reads them).  The
trainer (Train/trainer1.run_epoch) and bench.py switch it on; a caller that wants every logit leaves it off."""
from .. import ops
from .modules import get_src_mask, get_trg_mask


def _trg_mask(trg_in, pad_id, use_cond2dec, dconds=None):
    """get_trg_mask; on the GPU and without cond2dec its uint8 form comes from the token ids in one launch
    (ops.trg_mask_u8) instead of the int64 [B,T,T] tensor -- the model only asks whether an entry is nonzero."""
    if trg_in.is_cuda and not use_cond2dec and trg_in.dim() == 2 and trg_in.stride(1) == 1:
        return ops.trg_mask_u8(trg_in, pad_id)
    return get_trg_mask(trg_in, pad_id, use_cond2dec, dconds)


def _loss_rows(batch, pad_id, use_cond2dec, skip_ignored):
    if not skip_ignored or use_cond2dec:
        return None
    return batch["trg"][:, 1:] != pad_id


def _plain(model, batch, pad_id, use_cond2dec, skip_ignored=False):
    trg_in = batch["trg"][:, :-1]
    kw = {}
    rows = _loss_rows(batch, pad_id, use_cond2dec, skip_ignored)
    if rows is not None:
        kw["loss_rows"] = rows
    return model.forward(src=batch["src"], trg=trg_in,
                         src_mask=get_src_mask(batch["src"], pad_id),
                         trg_mask=_trg_mask(trg_in, pad_id, use_cond2dec), **kw)


def _conditioned(model, batch, pad_id, use_cond2dec, skip_ignored=False):
    trg_in = batch["trg"][:, :-1]
    kw = {}
    rows = _loss_rows(batch, pad_id, use_cond2dec, skip_ignored)
    if rows is not None:
        kw["loss_rows"] = rows
    return model.forward(src=batch["src"], trg=trg_in,
                         src_mask=get_src_mask(batch["src"], pad_id, batch["econds"]),
                         trg_mask=_trg_mask(trg_in, pad_id, use_cond2dec, batch["dconds"]),
                         econds=batch["econds"], dconds=batch["dconds"], **kw)


forward_propagation = {
    "vaetf": _plain,
    "scavaetf": _plain,
    "pvaetf": _conditioned,
    "pscavaetf": _conditioned,
}
