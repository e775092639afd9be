"""The training-harness surface of the reference, on the HIP training step.

`SpeakerDiarization` mirrors FS-EEND/train/oln_tfm_enc_dec.py:18-313 (and, with ``pit=True``, the label permutation of
train/oln_tfm_enc_dec_spk_pit.py:78-87): same constructor ``(hparams, model, datasets, opt, scheduler, collate_func)``
and the same LightningModule-style methods (`training_step`, `validation_step`, `validation_epoch_end`, `test_step`,
`test_epoch_end`, `predict_step`, `configure_optimizers`, `*_dataloader`).  pytorch_lightning is not a dependency:
`Trainer` below is the minimal loop that drives these methods the way Lightning 1.8 does (FS-EEND/train_dia.py:145-160):
one process per GPU, `strategy="ddp"` = one all-reduce of the flat gradient buffer per optimiser step over
torch.distributed (backend "nccl" = RCCL over xGMI), `gradient_clip_val`, Noam stepped per optimiser step.

`opt` / `scheduler` may be the objects the reference script builds (torch.optim.Adam and utlis/scheduler.NoamScheduler:
only their hyper-parameters are read -- the update itself is eend_adam_step_f32 on the flat buffers) or plain dicts.
"""
import math
from collections import defaultdict
from typing import List, Sequence

import torch
import torch.nn.functional as F
from torch import Tensor

from . import postproc
from .lib import EendHipError


def prepare_labels(labels: Sequence[Tensor], clip_lengths: Sequence[int]) -> List[Tensor]:
    """train/oln_tfm_enc_dec.py:53-75 (identical in validation_step / test_step): pad the speaker columns, order each
    utterance's speakers by first activity (never-active last), prepend the silence column, append the all-zero
    "none speaker" column, cut to (ilen, nspk_i + 2).  Index work on the (B, T, S) label tensor, on its own device."""
    n_spks = [l.shape[1] for l in labels]
    max_spk = max(n_spks)
    lab = [F.pad(l, (0, max_spk - l.shape[1])) for l in labels]
    lab = torch.nn.utils.rnn.pad_sequence(lab, padding_value=0.0, batch_first=True)
    B, T, _ = lab.shape
    frame_index = torch.arange(1, T + 1, device=lab.device, dtype=lab.dtype)[None, :, None]
    first = frame_index * lab
    first = first.masked_fill(first == 0, float("inf")).min(dim=1)[0]
    order = torch.argsort(first, dim=1, stable=True)        # ties (same first frame / never active) keep column order, as the CPU reference sorts them
    lab = torch.gather(lab, 2, order[:, None, :].expand(B, T, max_spk))
    silence = 1.0 - lab.max(dim=-1)[0]
    lab = torch.cat([silence[..., None], lab, torch.zeros(B, T, 1, dtype=lab.dtype, device=lab.device)], dim=-1)
    return [lab[b, :l, :n + 2] for b, (l, n) in enumerate(zip(clip_lengths, n_spks))]


def _hyper(opt, scheduler, hparams):
    """Optimiser hyper-parameters from whatever the caller built (FS-EEND/train_dia.py:77-100)."""
    tr = (hparams or {}).get("training", {}) if isinstance(hparams, dict) else {}
    h = dict(lr=float(tr.get("lr", 1.0)), betas=(0.9, 0.98), eps=1e-9, warmup=int(tr.get("warm_steps", 100000)),
             scale=float(tr.get("schedule_scale", 1.0)), clip=float(tr.get("grad_clip", 5.0) or 0.0), noam=True)
    if isinstance(opt, dict):
        h.update({k: opt[k] for k in ("lr", "betas", "eps") if k in opt})
    elif opt is not None and hasattr(opt, "param_groups"):
        g = opt.param_groups[0]
        h["lr"] = float(getattr(scheduler, "base_lrs", [g.get("initial_lr", g["lr"])])[0]) if scheduler is not None else float(g["lr"])
        h["betas"], h["eps"] = tuple(g.get("betas", h["betas"])), float(g.get("eps", h["eps"]))
    if isinstance(scheduler, dict):
        h["warmup"], h["scale"] = int(scheduler.get("warmup_steps", h["warmup"])), float(scheduler.get("scale", h["scale"]))
    elif scheduler is not None:
        h["warmup"], h["scale"] = int(getattr(scheduler, "warmup_steps", h["warmup"])), float(getattr(scheduler, "scale", h["scale"]))
    else:
        h["noam"] = False
    return h


class SpeakerDiarization:
    """LightningModule-shaped training / evaluation module for the FS-EEND mirror model."""

    def __init__(self, hparams, model, datasets, opt, scheduler, collate_func, pit: bool = False):
        self.hparams = dict(hparams)
        self.datasets, self.model, self.opt, self.scheduler, self.collate_func = datasets, model, opt, scheduler, collate_func
        self.max_spks = self.hparams["data"]["max_speakers"]
        self.label_delay = self.hparams["data"].get("label_delay", 0)
        if self.label_delay:
            raise NotImplementedError("label_delay != 0 is not used by any shipped FS-EEND config")
        self.pit = pit
        self._step = None
        self.logged = {}
        self.global_rank, self.world_size = 0, 1

    # ---- plumbing
    def log(self, key, value, **kw):
        self.logged[key] = value

    def _engine(self):
        """The native training step of the model family: FS-EEND (train.FsTrainStep) or LS-EEND (train_ls.LsTrainStep,
        with `training.sync_batchnorm` as LS-EEND/train_dia_simu.py:167 passes it to Lightning).  scheduler=None runs the
        optimiser at the constant configured lr, like the reference scripts do (train_dia.py:95-100)."""
        if self._step is None:
            from .ls_model import OnlineConformerRetentionDADiarization as LsModel
            h = _hyper(self.opt, self.scheduler, self.hparams)
            kw = dict(warmup=h["warmup"] if h["noam"] else None, lr=h["lr"], schedule_scale=h["scale"], grad_clip=h["clip"],
                      betas=h["betas"], eps=h["eps"], drop_seed=self._drop_seed())
            if isinstance(self.model, LsModel):
                from .train_ls import LsTrainStep
                sync = bool((self.hparams.get("training") or {}).get("sync_batchnorm", True))
                self._step = LsTrainStep(self.model, sync_batchnorm=sync, **kw)
            else:
                from .train import FsTrainStep
                self._step = FsTrainStep(self.model, **kw)
        return self._step

    def _drop_seed(self):
        """Seed of the dropout hash: the run's `training.seed` (train_dia.py seeds torch from it) offset by the rank, so
        data-parallel replicas draw different masks."""
        seed = int((self.hparams.get("training") or {}).get("seed", 0) or 0)
        rank = torch.distributed.get_rank() if torch.distributed.is_available() and torch.distributed.is_initialized() else 0
        return seed * 1000003 + rank

    def to(self, device):
        self.model.to(device)
        return self

    def state_dict(self):
        return {"model." + k: v for k, v in self.model.state_dict().items()}

    def load_state_dict(self, sd, strict=True):
        sd = {k[len("model."):] if k.startswith("model.") else k: v for k, v in sd.items()}
        with torch.no_grad():
            own = self.model.state_dict()
            missing = [k for k in own if k not in sd]
            if strict and (missing or [k for k in sd if k not in own]):
                raise KeyError(f"state_dict mismatch: missing {missing}")
            for k, v in sd.items():
                if k in own:
                    own[k].copy_(v)                    # in place: parameters may be views of the flat buffer
        self.model._prep = None
        if self._step is not None:
            self._step.prep_weights()

    # ---- LightningModule methods
    def detect(self, x, labels, ilens):
        return self.model(x, tgt=labels, ilens=ilens)

    def training_step(self, batch, batch_index):
        """forward (HIP) + losses; the gradients are produced by `backward()`, the update by `optimizer_step()`
        (Lightning calls them in this order; `Trainer` below does the same)."""
        feats, labels, _ = batch
        eng = self._engine()
        dev = eng.dev
        clip_lengths = [x.shape[0] for x in feats]
        labels = prepare_labels([l.to(dev, torch.float32) for l in labels], clip_lengths)
        if not getattr(eng, "_prepped", False):
            eng.prep_weights()
            eng._prepped = True
        self._bf = eng.forward(feats, labels, clip_lengths, pit=self.pit)
        pit_loss, emb_loss = self._bf.loss[0], self._bf.loss[1]
        tot_loss = pit_loss + emb_loss
        self.log("train/lr", getattr(eng, "last_lr", 0.0), prog_bar=True)
        self.log("train/pit_loss", pit_loss)
        self.log("train/emb_loss", emb_loss)
        self.log("train/tot_loss", tot_loss)
        return tot_loss

    def backward(self):
        self._engine().backward(self._bf)

    def optimizer_step(self):
        eng = self._engine()
        eng.all_reduce_grads()
        return eng.optimizer_step()

    @torch.no_grad()
    def validation_step(self, batch, batch_index):
        feats, labels, _ = batch
        dev = next(self.model.parameters()).device
        clip_lengths = [x.shape[0] for x in feats]
        n_spks = [l.shape[1] for l in labels]
        labels = prepare_labels([l.to(dev, torch.float32) for l in labels], clip_lengths)
        was = self.model.training
        self.model.eval()
        preds, emb_loss, _, _ = self.detect([f.to(dev) for f in feats], labels, clip_lengths)
        self.model.train(was)
        preds_realspk = [p[:, 1:-1] for p in preds]
        labels_realspk = [l[:, 1:-1] for l in labels]
        stats = postproc.report_diarization_error(preds_realspk, labels_realspk, label_delay=self.label_delay)
        self.log("val/emb_loss", emb_loss)
        return stats

    def _epoch_end(self, outputs, prefix):
        holder = defaultdict(list)
        for stats in outputs:
            for k, v in stats.items():
                holder[k] += v
        avg = {k: sum(v) / len(v) for k, v in holder.items()}
        avg["DER"] = avg["diarization_error"] / avg["speaker_scored"] if avg.get("speaker_scored") else float("nan")
        for k, v in avg.items():
            self.log(f"{prefix}/{k}", v, sync_dist=True)
        self.log(f"{prefix}/obj_metric", avg["DER"], sync_dist=True)
        return avg

    def validation_epoch_end(self, val_step_outputs):
        return self._epoch_end(val_step_outputs, "val")

    @torch.no_grad()
    def test_step(self, batch, batch_index):
        feats, labels, rec = batch
        dev = next(self.model.parameters()).device
        clip_lengths = [x.shape[0] for x in feats]
        n_spks = [l.shape[1] for l in labels]
        labels = prepare_labels([l.to(dev, torch.float32) for l in labels], clip_lengths)
        preds, embs, attractors = self.model.test([f.to(dev) for f in feats], clip_lengths, self.max_spks + 2)
        preds_realspk = [p[:, 1:nspk + 1] for p, nspk in zip(preds, n_spks)]
        labels_realspk = [l[:, 1:-1] for l in labels]
        return postproc.report_diarization_error(preds_realspk, labels_realspk, label_delay=self.label_delay)

    def test_epoch_end(self, test_step_outputs):
        avg = self._epoch_end(test_step_outputs, "test")
        for k in ("speaker_miss", "speaker_falarm", "speaker_error"):
            if avg.get("speaker_scored"):
                self.log(f"test/{k}_rate", avg[k] / avg["speaker_scored"])
        return avg

    @torch.no_grad()
    def predict_step(self, batch, batch_index):
        feats, rec = batch
        dev = next(self.model.parameters()).device
        clip_lengths = [x.shape[0] for x in feats]
        preds, _, _ = self.model.test([f.to(dev) for f in feats], clip_lengths, self.max_spks + 2)
        return preds[0][:, 1:]

    def configure_optimizers(self):
        if self.scheduler is not None:
            return {"optimizer": self.opt, "lr_scheduler": {"scheduler": self.scheduler, "interval": "step"}}
        return {"optimizer": self.opt}

    def _loader(self, split, batch_size, shuffle, sampler=None):
        from torch.utils.data import DataLoader
        tr = self.hparams.get("training", {})
        return DataLoader(self.datasets[split], batch_size=batch_size, shuffle=shuffle if sampler is None else False,
                          sampler=sampler, num_workers=tr.get("n_workers", 0), collate_fn=self.collate_func)

    def train_dataloader(self, sampler=None):
        tr = self.hparams["training"]
        return self._loader("train", tr["batch_size"], tr.get("shuffle", True), sampler)

    def val_dataloader(self):
        return self._loader("val", self.hparams["training"]["batch_size"], False)

    def test_dataloader(self):
        return self._loader("val", 1, False)

    def predict_dataloader(self):
        return self._loader("val", 1, False)


class EarlyStopping:
    """The other callback every reference training script passes (FS-EEND/train_dia.py:111-116, LS-EEND/train_dia_simu.py:127,
    train_dia_fintun_real.py): EarlyStopping(monitor="val/obj_metric", patience=early_stop_epoch, mode="min").  After each
    validation the freshly logged metric is compared with the best one so far; `patience` validations without an improvement
    of more than `min_delta` stop the fit loop (Lightning's semantics; the decision is taken on rank 0 and broadcast)."""

    def __init__(self, monitor="val/obj_metric", patience=3, mode="min", min_delta=0.0, verbose=False, **_lightning_only):
        if mode not in ("min", "max"):
            raise ValueError(f"EarlyStopping: mode must be 'min' or 'max', got {mode!r}")
        self.monitor, self.patience, self.mode, self.min_delta, self.verbose = monitor, int(patience), mode, abs(float(min_delta)), verbose
        self.best, self.wait, self.stopped_epoch = None, 0, None

    def on_validation_end(self, trainer, module, epoch):
        """-> True when training should stop."""
        value = module.logged.get(self.monitor)
        if value is None:
            raise RuntimeError(f"EarlyStopping: metric {self.monitor!r} was not logged by validation_epoch_end "
                               f"(logged: {sorted(module.logged)})")
        value = float(value)
        if value != value or value in (float("inf"), float("-inf")):      # Lightning's check_finite default: a NaN / inf metric stops the run
            self.stopped_epoch = epoch
            return True
        better = self.best is None or (value < self.best - self.min_delta if self.mode == "min" else value > self.best + self.min_delta)
        if better:
            self.best, self.wait = value, 0
            return False
        self.wait += 1
        if self.wait >= self.patience:
            self.stopped_epoch = epoch
            return True
        return False


    def state_dict(self):
        return dict(best=self.best, wait=self.wait, stopped_epoch=self.stopped_epoch)

    def load_state_dict(self, st):
        self.best, self.wait, self.stopped_epoch = st.get("best"), int(st.get("wait", 0)), st.get("stopped_epoch")


class ModelCheckpoint:
    """The checkpoint callback the reference passes to its Trainer (FS-EEND/train_dia.py:118-121: monitor val/obj_metric,
    save_top_k, save_last; LS-EEND/train_dia_simu.py:131).  Files are named like Lightning's default,
    `epoch={e}-step={global_step}.ckpt` -- the names `select_epoch_checkpoints` / train_dia.py:166-175 parse -- and hold
    {"state_dict": {"model.<key>": tensor}, "optimizer_state": ..., "epoch", "global_step"}: enough to resume.
    save_top_k = -1 keeps every epoch (what checkpoint averaging wants), k > 0 the k best by `monitor`, 0 none."""

    def __init__(self, dirpath, monitor=None, save_top_k=-1, mode="min", save_last=True, **_lightning_only):
        self.dirpath, self.monitor, self.save_top_k, self.mode, self.save_last = dirpath, monitor, save_top_k, mode, save_last
        self.kept = []                       # (score, path)
        self.best_model_path = ""

    def on_epoch_end(self, trainer, module, epoch):
        import os
        os.makedirs(self.dirpath, exist_ok=True)
        if self.save_top_k == 0:
            if self.save_last:
                torch.save(trainer.checkpoint(module, epoch), os.path.join(self.dirpath, "last.ckpt"))
            return
        path = os.path.join(self.dirpath, f"epoch={epoch}-step={trainer.global_step}.ckpt")
        # rank on a metric only if THIS epoch's validation logged it (a stale value of an earlier epoch, or the epoch number of
        # an epoch without validation, must not beat real scores): missing -> worst possible
        fresh = self.monitor is not None and getattr(trainer, "validated_epoch", None) == epoch and self.monitor in module.logged
        if self.monitor is None:
            score = float(epoch)
        elif fresh:
            score = float(module.logged[self.monitor])
            if self.mode == "max":
                score = -score
        else:
            score = float("inf")
        # bookkeeping first, files second: the callback state stored INSIDE this epoch's checkpoint (and last.ckpt) must already list this
        # epoch -- otherwise a resumed run never prunes the newest file and best_model_path lags one epoch
        self.kept.append((score, path))
        pruned = []
        if self.save_top_k > 0 and self.monitor:
            self.kept.sort(key=lambda t: t[0])
            pruned = [old for _, old in self.kept[self.save_top_k:]]
            self.kept = self.kept[:self.save_top_k]
        self.best_model_path = min(self.kept, key=lambda t: t[0])[1]
        ckpt = trainer.checkpoint(module, epoch)
        if path not in pruned:
            torch.save(ckpt, path)
        if self.save_last:
            torch.save(ckpt, os.path.join(self.dirpath, "last.ckpt"))
        for old in pruned:
            if os.path.exists(old):
                os.remove(old)

    def state_dict(self):
        return dict(kept=list(self.kept), best_model_path=self.best_model_path)

    def load_state_dict(self, st):
        self.kept = [tuple(t) for t in st.get("kept", [])]
        self.best_model_path = st.get("best_model_path", "")


class Trainer:
    """The slice of pytorch_lightning.Trainer the reference uses (FS-EEND/train_dia.py:145-160,185; LS-EEND/
    train_dia_simu.py:159-173), one process per GPU: max_epochs, callbacks (ModelCheckpoint, EarlyStopping above), strategy "ddp",
    sync_batchnorm (LS-EEND), accumulate_grad_batches, resume_from_checkpoint, gradient_clip_val,
    check_val_every_n_epoch, limit_*_batches.  Arguments it does not implement are refused, not ignored."""

    _NO_EFFECT = ("gpus", "devices", "accelerator", "logger", "profiler", "enable_progress_bar", "num_nodes", "deterministic")

    def __init__(self, max_epochs=1, gradient_clip_val=None, accumulate_grad_batches=1, check_val_every_n_epoch=1,
                 strategy=None, limit_train_batches=None, limit_val_batches=None, log_every_n_steps=100, callbacks=None,
                 num_sanity_val_steps=0, resume_from_checkpoint=None, sync_batchnorm=None, default_root_dir=None, **kw):
        unknown = [k for k in kw if k not in self._NO_EFFECT]
        if unknown:
            raise TypeError(f"Trainer: unsupported argument(s) {unknown} (not part of the reference's recipe)")
        if strategy not in (None, "ddp", "ddp_find_unused_parameters_false", "auto"):
            raise NotImplementedError(f"strategy={strategy!r}: one process per GPU with a flat-gradient all-reduce (ddp) only")
        self.accum = int(accumulate_grad_batches or 1)
        if self.accum < 1:
            raise ValueError("accumulate_grad_batches must be >= 1")
        self.callbacks = list(callbacks or [])
        for cb in self.callbacks:
            if not isinstance(cb, (ModelCheckpoint, EarlyStopping)):
                raise TypeError(f"Trainer: unsupported callback {type(cb).__name__} (fs_eend_amd.trainer.ModelCheckpoint / EarlyStopping only)")
        if default_root_dir and not any(isinstance(cb, ModelCheckpoint) for cb in self.callbacks):
            self.callbacks.append(ModelCheckpoint(default_root_dir))
        self.validated_epoch = None
        self.should_stop = False
        self.max_epochs, self.clip, self.val_every = max_epochs, gradient_clip_val, check_val_every_n_epoch
        self.limit_train, self.limit_val, self.log_every = limit_train_batches, limit_val_batches, log_every_n_steps
        self.strategy, self.resume, self.sync_bn = strategy, resume_from_checkpoint, sync_batchnorm
        self.global_step = 0
        self.current_epoch = 0
        self.history = []

    @property
    def checkpoint_callback(self):
        return next((cb for cb in self.callbacks if isinstance(cb, ModelCheckpoint)), None)

    @staticmethod
    def _dist():
        import torch.distributed as dist
        return dist if dist.is_available() and dist.is_initialized() else None

    def checkpoint(self, module, epoch):
        eng = module._engine()
        return dict(state_dict={k: v.detach().cpu().clone() for k, v in module.state_dict().items()},
                    optimizer_state={k: (v.cpu() if isinstance(v, Tensor) else v) for k, v in eng.optimizer_state().items()},
                    epoch=epoch, global_step=self.global_step,
                    callbacks={f"{i}:{type(cb).__name__}": cb.state_dict() for i, cb in enumerate(self.callbacks) if hasattr(cb, "state_dict")})

    def _restore(self, module, path):
        ckpt = torch.load(path, map_location="cpu")
        if "state_dict" not in ckpt:                     # a bare state dict (LS-EEND/train_dia_simu.py:196-199 saves those)
            module.load_state_dict(ckpt)
            return 0
        module.load_state_dict(ckpt["state_dict"])
        eng = module._engine()
        if ckpt.get("optimizer_state") is not None:
            eng.load_optimizer_state(ckpt["optimizer_state"])
        eng.prep_weights()
        self.global_step = int(ckpt.get("global_step", 0))
        for i, cb in enumerate(self.callbacks):          # top-k / patience bookkeeping of the run being resumed (keyed by position: two
            saved = ckpt.get("callbacks", {})            # callbacks of one class keep separate states; round-4 files used the bare class name)
            st = saved.get(f"{i}:{type(cb).__name__}", saved.get(type(cb).__name__))
            if st is not None and hasattr(cb, "load_state_dict"):
                cb.load_state_dict(st)
        return int(ckpt.get("epoch", -1)) + 1

    def fit(self, module: SpeakerDiarization):
        dist = self._dist()
        rank, world = (dist.get_rank(), dist.get_world_size()) if dist else (0, 1)
        module.global_rank, module.world_size = rank, world
        if self.clip is not None:
            module.hparams.setdefault("training", {})["grad_clip"] = self.clip
        if self.sync_bn is not None:
            module.hparams.setdefault("training", {})["sync_batchnorm"] = bool(self.sync_bn)
        eng = module._engine()
        first_epoch = self._restore(module, self.resume) if self.resume else 0
        if dist and world > 1:                         # DDP start-up: every rank begins from rank 0's parameters / buffers
            dist.broadcast(eng.flat.params, 0)
            for b in module.model.buffers():
                dist.broadcast(b, 0)
            eng.prep_weights()
        sampler = None
        if world > 1:
            from torch.utils.data.distributed import DistributedSampler
            sampler = DistributedSampler(module.datasets["train"], num_replicas=world, rank=rank,
                                         shuffle=module.hparams["training"].get("shuffle", True))
        for epoch in range(first_epoch, self.max_epochs):
            self.current_epoch = epoch
            if sampler is not None:
                sampler.set_epoch(epoch)
            module.model.train()
            micro = 0
            for bi, batch in enumerate(module.train_dataloader(sampler)):
                if self.limit_train is not None and bi >= self.limit_train:
                    break
                loss = module.training_step(batch, bi)
                module.backward()
                eng.accumulate_grads(micro, self.accum)      # no-op for accumulate_grad_batches == 1
                micro += 1
                if micro < self.accum:
                    continue
                micro = 0
                lr = module.optimizer_step()
                self.global_step += 1
                if self.global_step % self.log_every == 0 or self.global_step == 1:
                    self.history.append(dict(step=self.global_step, loss=float(loss), lr=lr))
            if micro:                                 # epoch length not a multiple of accumulate_grad_batches: step with what there is
                _finish_partial(eng, micro, self.accum)
                module.optimizer_step()
                self.global_step += 1
            if self.val_every and (epoch + 1) % self.val_every == 0 and "val" in module.datasets:
                outs = []
                module.model.eval()
                for bi, batch in enumerate(module.val_dataloader()):
                    if self.limit_val is not None and bi >= self.limit_val:
                        break
                    outs.append(module.validation_step(batch, bi))
                if outs:
                    module.validation_epoch_end(outs)
                    self.validated_epoch = epoch
                    stop = False
                    if rank == 0:
                        # every callback sees every validation (a generator inside any() would stop updating them at the first True)
                        stop = any([cb.on_validation_end(self, module, epoch) for cb in self.callbacks if isinstance(cb, EarlyStopping)])
                    if dist and world > 1:               # every rank leaves the loop in the same epoch
                        flag = torch.tensor([1 if stop else 0], device=eng.flat.params.device)
                        dist.broadcast(flag, 0)
                        stop = bool(int(flag.item()))
                    self.should_stop = stop
            if rank == 0:
                for cb in self.callbacks:
                    if isinstance(cb, ModelCheckpoint):
                        cb.on_epoch_end(self, module, epoch)
            if self.should_stop:
                break
        return self

    def test(self, module: SpeakerDiarization):
        outs = [module.test_step(b, i) for i, b in enumerate(module.test_dataloader())]
        return module.test_epoch_end(outs) if outs else {}


def _finish_partial(eng, done, count):
    """`done` < `count` micro-batches were accumulated (each scaled 1/count) when the epoch ended: move the running sum
    into the gradient buffer so the optimiser steps with it (Lightning steps on the last, shorter group too)."""
    from .train import _call
    _call("eend_grad_accumulate_f32", eng.flat.grads, eng._gacc, 1.0, 1, eng.flat.numel)


def average_checkpoints(state_dicts):
    """FS-EEND/train_dia.py:176-180 / utlis/avg_ckpt.py:6-22: element-wise mean of the listed state dicts
    (`test_state[name] += param / len(ckpts)`, integer buffers included, in the reference's accumulation order)."""
    out = defaultdict(float)
    n = len(state_dicts)
    for sd in state_dicts:
        for name, param in sd.items():
            out[name] += param / n
    return dict(out)
