"""torch.autograd bridge: `model(src, tgt, ilens)` with gradients enabled (what the reference's
`SpeakerDiarization.training_step` does, FS-EEND/train/oln_tfm_enc_dec.py:47-49,77) runs the HIP training forward and
returns tensors whose backward is the hand-written HIP backward (train.FsTrainStep.backward).  The caller's own loss
(e.g. the reference's `standard_loss(preds, labels) + emb_loss`, train/utils/loss.py:119-125) is ordinary torch on the
small (T_i, C) logit slices; `loss.backward()` hands d loss / d logits and d loss / d emb_loss to the kernels, which fill
`.grad` of every parameter.  Optimiser, gradient clipping and DDP hooks of the caller's framework work unchanged.

This is the drop-in path.  The native path (trainer.SpeakerDiarization / train.FsTrainStep.step) fuses loss, clipping and
Adam into the step and is what bench.py --mode train measures.
"""
import torch

from .lib import EendHipError


class _FsForward(torch.autograd.Function):
    @staticmethod
    def forward(ctx, holder, *params):
        eng, src, tgt, ilens = holder["eng"], holder["src"], holder["tgt"], holder["ilens"]
        eng.prep_weights()                                   # the caller's optimiser may have updated the parameters
        bf = eng.forward(src, tgt, ilens, fused_loss=False)
        ctx.eng, ctx.bf, ctx.n_params = eng, bf, len(params)
        emb = bf.emb32.view(bf.shape[0], bf.shape[2], 256)
        ctx.mark_non_differentiable(emb, bf.attr_n)
        return bf.logits_full, bf.loss[1].clone(), emb, bf.attr_n

    @staticmethod
    def backward(ctx, dlogits, demb_loss, _demb, _dattr):
        eng, bf = ctx.eng, ctx.bf
        if dlogits is None:
            dlogits = torch.zeros_like(bf.logits_full)
        coef = 0.0 if demb_loss is None else float(demb_loss)            # one host read: the drop-in path, not the fast path
        eng.backward(bf, dlogits=dlogits, emb_loss_grad=coef)
        fl = eng.flat
        grads = tuple(fl.g(n).clone() for n in fl.names)
        return (None,) + grads


def ls_forward_with_grad(model, src, tgt, ilens):
    """OnlineConformerRetentionDADiarization.forward (reference LS model :74-122) with autograd support: the same bridge
    over train_ls.LsTrainStep (train-mode BatchNorm statistics of the conv modules, SyncBatchNorm exchange when a process
    group is initialised)."""
    if not model.training:
        raise EendHipError("gradient-enabled forward needs model.train() (the conv modules' BatchNorm uses batch statistics); "
                           "use torch.no_grad() for evaluation")
    eng = getattr(model, "_autograd_engine", None)
    if eng is None:
        from .train_ls import LsTrainStep
        eng = LsTrainStep(model, drop_seed=torch.initial_seed() & 0xFFFFFFFF)
        object.__setattr__(model, "_autograd_engine", eng)
    params = [p for _, p in model.named_parameters()]
    holder = dict(eng=eng, src=src, tgt=tgt, ilens=ilens)
    logits, emb_loss, emb, attr = _FsForward.apply(holder, *params)
    n_speakers = [t.shape[1] for t in tgt]
    # the reference's head sees the length-masked embeddings (LS model :100,:116): rows beyond ilen are sliced off anyway
    output = [logits[b, :l, :n] for b, (l, n) in enumerate(zip(ilens, n_speakers))]
    embs = [emb[b, :l] for b, l in enumerate(ilens)]
    attractors = [attr[b, :l, 1:n] for b, (l, n) in enumerate(zip(ilens, n_speakers))]
    return output, emb_loss, embs, attractors


def fs_forward_with_grad(model, src, tgt, ilens):
    """OnlineTransformerDADiarization.forward (reference model :32-65) with autograd support."""
    if not model.training:
        raise EendHipError("gradient-enabled forward needs model.train() (BatchNorm batch statistics, FS model :165-166); "
                           "use torch.no_grad() for evaluation")
    eng = getattr(model, "_autograd_engine", None)
    if eng is None:
        from .train import FsTrainStep
        eng = FsTrainStep(model, drop_seed=torch.initial_seed() & 0xFFFFFFFF)   # dropout masks follow torch.manual_seed
        object.__setattr__(model, "_autograd_engine", eng)
    params = [p for _, p in model.named_parameters()]
    holder = dict(eng=eng, src=src, tgt=tgt, ilens=ilens)
    logits, emb_loss, emb, attr = _FsForward.apply(holder, *params)
    n_speakers = [t.shape[1] for t in tgt]
    output = [logits[b, :l, :n] for b, (l, n) in enumerate(zip(ilens, n_speakers))]
    embs = [emb[b, :l] for b, l in enumerate(ilens)]
    attractors = [attr[b, :l, 1:n] for b, (l, n) in enumerate(zip(ilens, n_speakers))]
    return output, emb_loss, embs, attractors
