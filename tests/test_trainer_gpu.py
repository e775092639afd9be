"""GPU: the Trainer surface around the native training step -- gradient accumulation (train_dia.py:151
accumulate_grad_batches), per-epoch checkpoints with Lightning's file names and bit-exact resume (train_dia.py:118-121,
:152 resume_from_checkpoint), the scheduler=None constant-lr path, refusal of unsupported arguments."""
import os

import pytest
import torch

from oracle import fixtures as FX

pytestmark = pytest.mark.gpu
CFG = dict(n_units=256, n_heads=4, enc_n_layers=1, dec_n_layers=1, dropout=0.0, has_mask=True, max_seqlen=500,
           dec_dim_feedforward=256, mask_delay=0)


class _Set(torch.utils.data.Dataset):
    def __init__(self, n, T, nspk, seed):
        self.f = FX.make_src([T] * n, 345, seed)
        self.l = FX.make_labels([T] * n, [nspk] * n, seed + 1)

    def __len__(self):
        return len(self.f)

    def __getitem__(self, i):
        return self.f[i], self.l[i], f"rec{i}"


def _collate(batch):
    f, l, r = zip(*batch)
    return list(f), list(l), list(r)


def _module(dev, scheduler=True, seed=3):
    from fs_eend_amd.fs_model import OnlineTransformerDADiarization
    from fs_eend_amd.trainer import SpeakerDiarization
    torch.manual_seed(seed)
    m = OnlineTransformerDADiarization(n_speakers=None, in_size=345, **CFG).to(dev).train()
    hp = dict(data=dict(max_speakers=3, label_delay=0), training=dict(lr=1.0 if scheduler else 1e-4, warm_steps=50, schedule_scale=1.0,
                                                                      grad_clip=5.0, batch_size=2, shuffle=False, seed=5))
    ds = {"train": _Set(8, 96, 2, 11)}
    opt = dict(lr=hp["training"]["lr"], betas=(0.9, 0.98), eps=1e-9)
    return SpeakerDiarization(hp, m, ds, opt, dict(warmup_steps=50, scale=1.0) if scheduler else None, _collate)


def test_accumulated_gradient_is_the_mean_of_the_micro_batches(hip_lib, dev):
    mod = _module(dev)
    eng = mod._engine()
    eng.prep_weights()
    batches = list(mod.train_dataloader())[:2]
    gs = []
    for i, b in enumerate(batches):
        mod.training_step(b, i)
        mod.backward()
        gs.append(eng.flat.grads.clone())
        eng.accumulate_grads(i, 2)
    torch.cuda.synchronize()
    want = 0.5 * gs[0] + 0.5 * gs[1]
    assert torch.equal(eng.flat.grads, want)


def test_checkpoint_names_and_bit_exact_resume(hip_lib, dev, tmp_path):
    from fs_eend_amd.trainer import ModelCheckpoint, Trainer
    d = str(tmp_path / "ck")
    a = _module(dev)
    Trainer(max_epochs=3, accumulate_grad_batches=2, callbacks=[ModelCheckpoint(d, save_top_k=-1)]).fit(a)
    torch.cuda.synchronize()
    files = sorted(os.listdir(d))
    assert files == ["epoch=0-step=2.ckpt", "epoch=1-step=4.ckpt", "epoch=2-step=6.ckpt", "last.ckpt"], files
    ck = torch.load(os.path.join(d, "epoch=1-step=4.ckpt"))
    assert all(k.startswith("model.") for k in ck["state_dict"]) and ck["optimizer_state"]["opt_step"] == 4
    # resume after epoch 1 into a fresh module: the third epoch must reproduce run `a` bit for bit
    b = _module(dev, seed=99)                              # different init: everything must come from the checkpoint
    tb = Trainer(max_epochs=3, accumulate_grad_batches=2, resume_from_checkpoint=os.path.join(d, "epoch=1-step=4.ckpt"))
    tb.fit(b)
    torch.cuda.synchronize()
    assert tb.global_step == 6
    assert torch.equal(a._engine().flat.params, b._engine().flat.params)
    assert torch.equal(a._engine().flat.m, b._engine().flat.m) and torch.equal(a._engine().flat.v, b._engine().flat.v)
    for (k, x), (_, y) in zip(a.model.state_dict().items(), b.model.state_dict().items()):
        assert torch.equal(x, y), k
    # the averaging recipe of train_dia.py:166-184 works on these files
    from fs_eend_amd.trainer import average_checkpoints
    avg = average_checkpoints([torch.load(os.path.join(d, f))["state_dict"] for f in files[:3]])
    assert set(avg) == set(ck["state_dict"])


def test_no_scheduler_means_constant_lr(hip_lib, dev):
    mod = _module(dev, scheduler=False)
    eng = mod._engine()
    assert eng.warmup is None
    lrs = []
    for i, b in enumerate(list(mod.train_dataloader())[:3]):
        mod.training_step(b, i)
        mod.backward()
        lrs.append(mod.optimizer_step())
    assert lrs == [1e-4, 1e-4, 1e-4]


def test_queued_steps_keep_their_own_hyperparameters(hip_lib, dev):
    """the lr / bias corrections of step k must be the ones step k's Adam kernel reads even when the host
    queues many steps ahead of the GPU -- two engines, one synchronising after every step, must agree bit for bit."""
    outs = []
    for sync in (True, False):
        mod = _module(dev)
        eng = mod._engine()
        eng.prep_weights()
        b = list(mod.train_dataloader())
        for i in range(24):
            bb = b[i % len(b)]
            mod.training_step(bb, i)
            mod.backward()
            mod.optimizer_step()
            if sync:
                torch.cuda.synchronize()
        torch.cuda.synchronize()
        outs.append(eng.flat.params.clone())
    assert torch.equal(outs[0], outs[1])


def test_unsupported_trainer_arguments_are_refused(hip_lib, dev):
    from fs_eend_amd.trainer import Trainer
    with pytest.raises(TypeError):
        Trainer(max_epochs=1, fast_dev_run=True)
    with pytest.raises(TypeError):
        Trainer(max_epochs=1, callbacks=[object()])
    with pytest.raises(NotImplementedError):
        Trainer(max_epochs=1, strategy="dp")
    Trainer(max_epochs=1, gpus=1, logger=None)            # accepted: no effect in a one-process-per-GPU world
