"""CPU: the two callbacks of the reference's training scripts (train_dia.py:111-121) as the Trainer implements them --
EarlyStopping's patience logic and ModelCheckpoint's ranking on freshly validated epochs only."""
import os
import types

import pytest
import torch


def _module(logged):
    return types.SimpleNamespace(logged=dict(logged))


def test_early_stopping_patience_and_modes():
    from fs_eend_amd.trainer import EarlyStopping
    es = EarlyStopping(monitor="val/obj_metric", patience=2, mode="min")
    vals = [0.5, 0.4, 0.41, 0.45, 0.3]
    stops = [es.on_validation_end(None, _module({"val/obj_metric": v}), e) for e, v in enumerate(vals)]
    assert stops == [False, False, False, True, False] and es.stopped_epoch == 3 and es.best == 0.3
    es = EarlyStopping(monitor="acc", patience=1, mode="max", min_delta=0.05)
    assert [es.on_validation_end(None, _module({"acc": v}), e) for e, v in enumerate([0.5, 0.54, 0.6])] == [False, True, False]
    with pytest.raises(RuntimeError):
        EarlyStopping(monitor="nope").on_validation_end(None, _module({"x": 1.0}), 0)
    with pytest.raises(ValueError):
        EarlyStopping(mode="sideways")


def test_trainer_accepts_the_reference_callbacks():
    from fs_eend_amd.trainer import EarlyStopping, ModelCheckpoint, Trainer
    t = Trainer(max_epochs=2, callbacks=[EarlyStopping(monitor="val/obj_metric", patience=10, mode="min"),
                                         ModelCheckpoint("/tmp/x", monitor="val/obj_metric", save_top_k=3, mode="min")])
    assert isinstance(t.checkpoint_callback, ModelCheckpoint)
    with pytest.raises(TypeError):
        Trainer(max_epochs=1, callbacks=[object()])


def test_checkpoint_ranks_only_fresh_validations(tmp_path):
    from fs_eend_amd.trainer import ModelCheckpoint
    cb = ModelCheckpoint(str(tmp_path), monitor="val/obj_metric", save_top_k=2, mode="min", save_last=False)
    trainer = types.SimpleNamespace(global_step=0, validated_epoch=None, checkpoint=lambda m, e: {"epoch": e})
    m = _module({})
    # epoch 0: no validation yet (check_val_every_n_epoch = 2) -> must not rank as score 0.0
    cb.on_epoch_end(trainer, m, 0)
    trainer.validated_epoch, m.logged["val/obj_metric"] = 1, 0.30
    cb.on_epoch_end(trainer, m, 1)
    cb.on_epoch_end(trainer, m, 2)                      # stale metric of epoch 1: not fresh
    trainer.validated_epoch, m.logged["val/obj_metric"] = 3, 0.25
    cb.on_epoch_end(trainer, m, 3)
    assert os.path.basename(cb.best_model_path) == "epoch=3-step=0.ckpt"
    assert sorted(os.path.basename(p) for _, p in cb.kept) == ["epoch=1-step=0.ckpt", "epoch=3-step=0.ckpt"]
    assert sorted(os.listdir(tmp_path)) == ["epoch=1-step=0.ckpt", "epoch=3-step=0.ckpt"]
    st = cb.state_dict()
    cb2 = ModelCheckpoint(str(tmp_path), monitor="val/obj_metric", save_top_k=2)
    cb2.load_state_dict(st)
    assert cb2.best_model_path == cb.best_model_path and len(cb2.kept) == 2


def test_checkpoint_file_carries_its_own_epoch_in_the_callback_state(tmp_path):
    """the callback state stored inside each .ckpt (and last.ckpt) must already list the epoch being saved, otherwise a
    resumed run never prunes the newest file (top-k leaves k + 1) and best_model_path lags by one epoch."""
    import torch
    from fs_eend_amd.trainer import ModelCheckpoint
    cb = ModelCheckpoint(str(tmp_path), monitor="val/obj_metric", save_top_k=1, mode="min", save_last=True)
    trainer = types.SimpleNamespace(global_step=0, validated_epoch=None)
    trainer.checkpoint = lambda m, e: {"epoch": e, "callbacks": {"0:ModelCheckpoint": cb.state_dict()}}
    m = _module({})
    for epoch, score in enumerate((0.5, 0.4, 0.45)):
        trainer.validated_epoch, m.logged["val/obj_metric"] = epoch, score
        cb.on_epoch_end(trainer, m, epoch)
        last = torch.load(os.path.join(tmp_path, "last.ckpt"))
        st = last["callbacks"]["0:ModelCheckpoint"]
        assert os.path.basename(st["best_model_path"]) == os.path.basename(cb.best_model_path)
        assert [os.path.basename(p) for _, p in st["kept"]] == [os.path.basename(p) for _, p in cb.kept]
    assert sorted(os.listdir(tmp_path)) == ["epoch=1-step=0.ckpt", "last.ckpt"]          # k = 1: the best one, plus last
    assert os.path.basename(cb.best_model_path) == "epoch=1-step=0.ckpt"


def test_early_stopping_state_and_non_finite_metric():
    from fs_eend_amd.trainer import EarlyStopping
    es = EarlyStopping(monitor="m", patience=3, mode="min")
    assert not es.on_validation_end(None, _module({"m": 1.0}), 0)
    assert not es.on_validation_end(None, _module({"m": 1.1}), 1)
    es2 = EarlyStopping(monitor="m", patience=3, mode="min")
    es2.load_state_dict(es.state_dict())                     # patience survives resume_from_checkpoint
    assert (es2.best, es2.wait) == (1.0, 1)
    assert not es2.on_validation_end(None, _module({"m": 1.2}), 2)
    assert es2.on_validation_end(None, _module({"m": 1.3}), 3)
    assert EarlyStopping(monitor="m", patience=100).on_validation_end(None, _module({"m": float("nan")}), 0)      # check_finite
