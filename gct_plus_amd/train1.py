"""Process bootstrap of the training job (reference train1.py:32-171).

One process per GPU.  Launch either like the reference scripts do --
`python train1.py -flags` (spawns one worker per visible GPU, Bashscript/train/*.sh) or
`torchrun ... train1.py -flags` -- or under `python -m torch.distributed.run` with
RANK/LOCAL_RANK/WORLD_SIZE set.  Differences from the reference, all forced by the build
environment and none touching the arithmetic:
  * torch DDP -> gct_plus_amd.dp.FlatDataParallel (RCCL all-reduce of the flat gradient buffer);
  * torch.optim.Adam -> gct_plus_amd.optim.FusedAdam (same state_dict layout);
  * data: `{prepared_folder}/train[_sca].csv` / `test[_sca].csv` (the reference's prepared CSVs,
    columns src[, src_scaffold], src_<prop>, trg_<prop>) go through the native tokeniser/collate
    of gct_plus_amd.data (replaces torchtext Field pickles; vocabularies are built from the
    training split and stored as {util_folder}/SRC[_sep].json, TRG[_sep].json); `-synthetic N`
    trains on synthetic MOSES-shaped token batches instead (no dataset can be downloaded here).
    SMILES randomisation (-randomize_prob, needs rdkit) is not available.
"""
import argparse
import logging
import os
import random
import sys

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from . import synthetic
from .Configuration.config import train_opts
from .Model.build_model import get_model
from .Train.trainer1 import train_model
from .dp import FlatDataParallel
from .optim import FusedAdam


def set_seed(seed):
    """reference Utils/seed.py:7-18"""
    if seed is None:
        return
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.manual_seed_all(seed)


def get_logger(name, log_path):
    """console + {model_folder}/records.log (reference Utils/log.py:26-43)"""
    log = logging.getLogger(name)
    log.setLevel(logging.INFO)
    if not log.handlers:
        fmt = logging.Formatter("%(asctime)s - %(levelname)s - %(message)s")
        for h in (logging.StreamHandler(sys.stdout), logging.FileHandler(log_path)):
            h.setFormatter(fmt)
            log.addHandler(h)
    return log


class ShardedLoader:
    """Minimal DataLoader+DistributedSampler stand-in over an in-HBM token dataset:
    per-rank index shard (synthetic.shard_indices == DistributedSampler semantics), batches
    of `batch_size` PER RANK (reference train1.py:83, Utils/dataset.py:304-329)."""

    def __init__(self, data, batch_size, rank, world, shuffle, seed, device):
        self.data = {k: v.to(device) for k, v in data.items()}
        self.bs, self.rank, self.world, self.shuffle, self.seed = batch_size, rank, world, shuffle, seed
        self.epoch = 0
        self.n = next(iter(data.values())).size(0)

    def set_epoch(self, epoch):
        self.epoch = epoch

    def __len__(self):
        per_rank = -(-self.n // self.world)
        return -(-per_rank // self.bs)

    def __iter__(self):
        idx = synthetic.shard_indices(self.n, self.world, self.rank, self.epoch, self.seed or 0,
                                      self.shuffle)
        idx = torch.tensor(idx, device=next(iter(self.data.values())).device)
        for s in range(0, idx.numel(), self.bs):
            sel = idx[s:s + self.bs]
            yield {k: v.index_select(0, sel) for k, v in self.data.items()}


def load_tokens(args, split, n_synth):
    if n_synth > 0:
        return synthetic.make_dataset(n_synth, args.max_strlen, args.model_type,
                                      seed=0 if split == "train" else 1)
    path = os.path.join(args.prepared_folder, f"{split}_tokens.pt")
    if not os.path.exists(path):
        raise FileNotFoundError(
            f"{path} not found. The torchtext/rdkit tokeniser of the reference is outside this "
            "build's scope; pass -synthetic N or provide pre-tokenised tensors.")
    return torch.load(path, weights_only=True)


def main(rank, world_size, argv=None):
    parser = argparse.ArgumentParser()
    train_opts(parser)
    args = parser.parse_args(argv)
    local = int(os.environ.get("LOCAL_RANK", rank))
    share_gpu = os.environ.get("GCT_DP_SHARE_GPU", "0") != "0"      # TEST RIG (one-GPU boxes): every rank on cuda:0,
    if share_gpu:                                                   # gradients through host memory over gloo
        local = 0
    torch.cuda.set_device(local)
    device = torch.device("cuda", local)
    if world_size > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share_gpu:
            dist.init_process_group(backend="gloo", rank=rank, world_size=world_size)
        else:
            dist.init_process_group(backend="nccl", rank=rank, world_size=world_size, device_id=device)
    set_seed(args.seed)
    os.makedirs(args.model_folder, exist_ok=True)
    LOG = get_logger("train", os.path.join(args.model_folder, "records.log"))
    LOG.info("random seed: %s", args.seed)
    if rank == 0:
        LOG.info(args)
        LOG.info(f"world size: {world_size}")
    if args.debug:
        args.batch_size = 4
    nc = len(args.property_list)
    sca = args.model_type in ("scavaetf", "pscavaetf")
    csv_tr = os.path.join(args.prepared_folder, "train_sca.csv" if sca else "train.csv")
    csv_va = os.path.join(args.prepared_folder, "test_sca.csv" if sca else "test.csv")
    if args.synthetic == 0 and os.path.exists(csv_tr):
        # real data: the reference's prepared CSVs (train1.py:68-73) through the native tokeniser
        import pandas as pd
        from . import data
        ftr, fva = pd.read_csv(csv_tr), pd.read_csv(csv_va)
        if args.debug:
            ftr, fva = ftr[:32], fva[:32]
        col = ((ftr["src_scaffold"].astype(str) + "<sep>" + ftr["src"].astype(str)) if sca
               else ftr["src"].astype(str)).tolist()
        SRC, TRG, _ = data.get_fields(args.model_type, args.util_folder, col)
        LOG.info(f"SRC: {SRC.stoi}")
        LOG.info(f"TRG: {TRG.stoi}")
        mk = lambda fr, sh: data.SmilesLoader(fr, SRC, TRG, args.model_type, args.property_list,
                                              args.batch_size, rank, world_size, sh, args.seed, device,
                                              args.use_scaffold)
        train_loader, valid_loader = mk(ftr, True), mk(fva, False)
        src_vocab, trg_vocab = len(SRC), len(TRG)
        args.sos_id, args.eos_id, args.pad_id = TRG.stoi["<sos>"], TRG.stoi["<eos>"], SRC.stoi["<pad>"]
    else:
        n_tr = 32 if (args.debug and args.synthetic) else args.synthetic
        n_va = 32 if (args.debug and args.synthetic) else (args.synthetic_valid or max(args.synthetic // 10, 0))
        train = load_tokens(args, "train", n_tr)
        valid = load_tokens(args, "test", n_va)
        if nc != synthetic.n_conds(args.model_type) and args.synthetic:
            raise ValueError(f"-property_list has {nc} entries but {args.model_type} expects "
                             f"{synthetic.n_conds(args.model_type)}")
        train_loader = ShardedLoader(train, args.batch_size, rank, world_size, True, args.seed, device)
        valid_loader = ShardedLoader(valid, args.batch_size, rank, world_size, False, args.seed, device)
        src_vocab, trg_vocab = synthetic.vocab_sizes(args.model_type)
        args.sos_id, args.eos_id, args.pad_id = synthetic.SOS_ID, synthetic.EOS_ID, synthetic.PAD_ID
    if rank == 0:
        LOG.info(f"# train / validation loader: {len(train_loader)} / {len(valid_loader)}")
    if args.start_epoch > 1:
        args.model_path = os.path.join(args.model_folder, f"model_{args.start_epoch-1}.pt")
    model = get_model(args, src_vocab, trg_vocab, rank).cuda()
    (model.sampler if hasattr(model, "sampler") else model.encoder).eps_mode = args.eps_mode
    total = sum(p.numel() for p in model.parameters())
    trainable = sum(p.numel() for p in model.parameters() if p.requires_grad)
    assert trainable > 0, "# trainable parameters = 0"
    LOG.info(f"# total params: {total}, # train params: {trainable}")
    inner = model
    if world_size > 1:
        kw = {}
        if share_gpu:        # RCCL refuses two ranks on one device: the rig stages the two collectives through the host
            from .testing import host_staged_allreduce, host_staged_broadcast
            kw = dict(allreduce=host_staged_allreduce, broadcast=host_staged_broadcast)
        model = FlatDataParallel(model, **kw)
    optimizer = FusedAdam(inner.parameters(), lr=args.lr, betas=(args.lr_beta1, args.lr_beta2),
                          eps=args.lr_eps, model=inner)
    if args.start_epoch > 1:
        from .Model.build_model import load_checkpoint
        ck = load_checkpoint(args.model_path)
        optimizer.load_state_dict(ck["opt_state_dict"])
    train_model(args, model, optimizer, train_loader, valid_loader, rank, world_size, LOG)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def _worker(rank, world_size, argv):
    os.environ["RANK"], os.environ["LOCAL_RANK"] = str(rank), str(rank)
    os.environ["WORLD_SIZE"] = str(world_size)
    main(rank, world_size, argv)


def cli(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    if "RANK" in os.environ and "WORLD_SIZE" in os.environ and int(os.environ["WORLD_SIZE"]) > 1:
        main(int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), argv)   # torchrun worker
        return
    n = torch.cuda.device_count()
    print("device count:", n)
    if os.environ.get("GCT_DP_RANKS"):          # test rigs: this many workers whatever the device count
        n = int(os.environ["GCT_DP_RANKS"])
    if n > 1:                                   # reference train1.py:156-166: one proc per GPU
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        mp.spawn(_worker, args=(n, argv), nprocs=n, join=True)
    else:
        main(0, 1, argv)
