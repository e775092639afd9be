"""GPU: the HIP LS-EEND forward against the golden vectors produced by the reference and
against the fp32 oracle.  Tolerance: logits within 1e-3 of the reference fp32 forward
(north_star); unit-vector outputs (emb / attractors) within 3e-3 per component."""
import pytest
import torch

from oracle import fixtures as FX
from oracle import ls_eend_ref as R
from tests.helpers import build_ls_mirror, ls_kwargs, max_abs

pytestmark = pytest.mark.gpu
LOGIT_TOL = 1e-3
VEC_TOL = 3e-3
LS_TEST = [c for c in FX.list_cases("ls_") if FX.load_case(c)[0]["kind"] == "ls_test"]


@pytest.mark.parametrize("name", LS_TEST)
def test_ls_test_vs_golden(hip_lib, dev, name):
    meta, arr = FX.load_case(name)
    m = build_ls_mirror(meta).to(dev)
    src = [s.to(dev) for s in FX.make_src(meta["lengths"], meta["in_size"], meta["xseed"])]
    got = m.test(src, meta["lengths"], max_nspks=meta["C"])
    torch.cuda.synchronize()
    r = meta["rows"]
    worst = 0.0
    for i in range(len(src)):
        assert got[0][i].shape == arr[f"logits{i}"].shape
        assert torch.isfinite(got[0][i]).all()
        worst = max(worst, max_abs(got[0][i], arr[f"logits{i}"]))
    print(f"{name}: max |logits - reference| = {worst:.2e}")
    # secondary, scale-free bar: random-init logits are cosines with std ~0.05, so the absolute 1e-3 bar is ~2 % of a
    # standard deviation; the relative RMS error over all frames must also stay below 1 %
    num = sum(float(((got[0][i].cpu().double() - torch.as_tensor(arr[f"logits{i}"]).double()) ** 2).sum()) for i in range(len(src)))
    den = sum(float((torch.as_tensor(arr[f"logits{i}"]).double() ** 2).sum()) for i in range(len(src)))
    rel_rms = (num / den) ** 0.5
    print(f"{name}: relative RMS logit error = {rel_rms:.2e}")
    assert rel_rms < 1e-2
    for i in range(len(src)):
        assert max_abs(got[0][i], arr[f"logits{i}"]) < LOGIT_TOL, f"{name}[{i}] logits {worst:.2e}"
        assert max_abs(got[1][i][::r], arr[f"emb{i}"]) < VEC_TOL
        assert max_abs(got[2][i][::r], arr[f"attr{i}"]) < VEC_TOL


def test_ls_stagewise_vs_oracle(hip_lib, dev):
    """Localise errors: embeddings (encoder + conv) and logits against the oracle on a small case."""
    meta, _ = FX.load_case("ls_chunk64_T200")
    m = build_ls_mirror(meta)
    src = FX.make_src(meta["lengths"], meta["in_size"], meta["xseed"])
    with torch.no_grad():
        want = R.ls_test(src, meta["lengths"], m.state_dict(), max_nspks=meta["C"], **ls_kwargs(meta))
    got = m.to(dev).test([s.to(dev) for s in src], meta["lengths"], max_nspks=meta["C"])
    e_emb = max_abs(got[1][0], want[1][0])
    e_lg = max_abs(got[0][0], want[0][0])
    print(f"emb err {e_emb:.2e} logits err {e_lg:.2e}")
    assert e_emb < VEC_TOL and e_lg < LOGIT_TOL


def test_ls_forward_vs_golden(hip_lib, dev):
    meta, arr = FX.load_case("ls_fwd_train")
    m = build_ls_mirror(meta).to(dev)
    src = [s.to(dev) for s in FX.make_src(meta["lengths"], meta["in_size"], meta["xseed"])]
    tgt = [t.to(dev) for t in FX.make_labels(meta["lengths"], meta["ncols"], meta["lseed"])]
    with torch.no_grad():
        logits, loss, emb, attr = m(src, tgt, meta["lengths"])
    assert abs(float(loss) - float(arr["emb_loss"][0])) < 2e-4
    for i in range(len(src)):
        assert logits[i].shape == arr[f"logits{i}"].shape
        assert max_abs(logits[i], arr[f"logits{i}"]) < LOGIT_TOL
        assert max_abs(attr[i][::meta["rows"]], arr[f"attr{i}"]) < VEC_TOL


def test_ls_batch_independence_and_causality(hip_lib, dev):
    meta, _ = FX.load_case("ls_T500_c3")
    m = build_ls_mirror(meta).to(dev)
    T = 700                                             # 2 chunks of 500, second partial
    src = [s.to(dev) for s in FX.make_src([T] * 4, 345, 4321)]
    out = m.test(src, [T] * 4, 4)
    sub = m.test(src[1:3], [T, T], 4)
    assert torch.equal(sub[0][0], out[0][1]) and torch.equal(sub[0][1], out[0][2])
    t0 = 560
    s2 = [s.clone() for s in src[:1]]
    s2[0][t0:] = s2[0][t0:] * -0.5 + 1.0
    out2 = m.test(s2, [T], 4)
    assert torch.equal(out2[0][0][: t0 - 9], m.test(src[:1], [T], 4)[0][0][: t0 - 9])


def test_ls_long_sequence_vs_oracle(hip_lib, dev):
    """Many carried chunks (the long-form streaming mechanism, BASELINE config 5): T = 5000 = 10 chunks of 500
    with the retention state scanned across all of them, against the fp32 oracle."""
    meta, _ = FX.load_case("ls_T500_c3")
    m = build_ls_mirror(meta)
    T = 5000
    src = FX.make_src([T], meta["in_size"], 9876)
    with torch.no_grad():
        want = R.ls_test(src, [T], m.state_dict(), max_nspks=4, **ls_kwargs(meta))
    got = m.to(dev).test([s.to(dev) for s in src], [T], max_nspks=4)
    err = max_abs(got[0][0], want[0][0])
    tail = max_abs(got[0][0][-500:], want[0][0][-500:])
    print(f"T=5000: max |logits - oracle| = {err:.2e} (last chunk {tail:.2e})")
    assert torch.isfinite(got[0][0]).all()
    assert err < LOGIT_TOL
