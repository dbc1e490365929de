"""model_type -> forward function (reference Model/forward_propagation1.py:4-48): builds the
masks from the batch (on the batch's device, no host round trip; the decoder's self-attention mask straight from
the token ids) and calls model.forward by keyword exactly like the reference.

skip_ignored (an extension, default False = the reference's behaviour): tell the model which decoder rows the
reference's loss looks at -- `ys = trg[:, 1:] != pad` (Train/trainer1.py:21-22,97: cross-entropy with
ignore_index = pad) -- so that it does not compute the others (56 % of the decoder rows at MOSES-like lengths).  Loss
and every gradient are unchanged; the logits of the ignored rows are not the reference's (nothing reads them).  The
trainer (Train/trainer1.run_epoch) and bench.py switch it on; a caller that wants every logit leaves it off."""
import os

from .. import ops
from .modules import get_src_mask, get_trg_mask

PLAN_AHEAD = os.environ.get("GCT_PLAN_AHEAD", "1") != "0"      # A/B switch of prefetch()


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


def _masks(batch, pad_id, use_cond2dec, skip_ignored, conditioned):
    trg_in = batch["trg"][:, :-1]
    rows = _loss_rows(batch, pad_id, use_cond2dec, skip_ignored)
    if conditioned:
        return (trg_in, get_src_mask(batch["src"], pad_id, batch["econds"]),
                _trg_mask(trg_in, pad_id, use_cond2dec, batch["dconds"]), rows)
    return trg_in, get_src_mask(batch["src"], pad_id), _trg_mask(trg_in, pad_id, use_cond2dec), rows


def _key(batch, pad_id, use_cond2dec, skip_ignored):
    s, t = batch["src"], batch["trg"]
    return (s.data_ptr(), t.data_ptr(), tuple(s.shape), tuple(t.shape), s._version, t._version, pad_id,
            bool(use_cond2dec), bool(skip_ignored))


def prefetch(model_type, model, batch, pad_id, use_cond2dec, skip_ignored=False):
    """Queue, NOW, the masks and the row maps of a batch that the NEXT call of forward_propagation[model_type] will get
    (same arguments).  The trainer calls it between a step's forward and its backward: the maps' one device->host
    read-back then completes while that backward runs, and the next step's forward is queued without the host waiting
    for the device (engine.RowPlan.launch).  Purely an optimisation: a batch that was not announced, or was modified
    since, is handled exactly as before.  GPU batches and models with plan_ahead only; a no-op otherwise."""
    inner = getattr(model, "module", model)
    if not (PLAN_AHEAD and hasattr(inner, "plan_ahead") and batch["src"].is_cuda):
        return
    conditioned = forward_propagation[model_type] is _conditioned
    trg_in, src_mask, trg_mask, rows = _masks(batch, pad_id, use_cond2dec, skip_ignored, conditioned)
    inner._gct_ahead = (_key(batch, pad_id, use_cond2dec, skip_ignored), src_mask, trg_mask, rows,
                        inner.plan_ahead(src_mask, trg_mask, rows, trg_in))


def _ahead(model, batch, pad_id, use_cond2dec, skip_ignored):
    inner = getattr(model, "module", model)
    got = getattr(inner, "_gct_ahead", None)
    if got is None:
        return None
    inner._gct_ahead = None
    return got[1:] if got[0] == _key(batch, pad_id, use_cond2dec, skip_ignored) else None


def _call(model, batch, pad_id, use_cond2dec, skip_ignored, conditioned):
    ahead = _ahead(model, batch, pad_id, use_cond2dec, skip_ignored)
    kw = {}
    if ahead is not None:
        src_mask, trg_mask, rows, kw["_plan_ahead"] = ahead
        trg_in = batch["trg"][:, :-1]
    else:
        trg_in, src_mask, trg_mask, rows = _masks(batch, pad_id, use_cond2dec, skip_ignored, conditioned)
    if rows is not None:
        kw["loss_rows"] = rows
    if conditioned:
        kw.update(econds=batch["econds"], dconds=batch["dconds"])
    return model.forward(src=batch["src"], trg=trg_in, src_mask=src_mask, trg_mask=trg_mask, **kw)


def _plain(model, batch, pad_id, use_cond2dec, skip_ignored=False):
    return _call(model, batch, pad_id, use_cond2dec, skip_ignored, False)


def _conditioned(model, batch, pad_id, use_cond2dec, skip_ignored=False):
    return _call(model, batch, pad_id, use_cond2dec, skip_ignored, True)


forward_propagation = {
    "vaetf": _plain,
    "scavaetf": _plain,
    "pvaetf": _conditioned,
    "pscavaetf": _conditioned,
}
