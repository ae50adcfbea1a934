"""Host logic of optim.FusedAdamW that needs no GPU: torch-compatible construction / state_dict layout, loud failure on
CPU parameters (no CPU fallback), launch-cache hygiene."""
import pytest
import torch

from uni_renderer_amd.optim import FusedAdamW


def _params():
    g = torch.Generator().manual_seed(0)
    return [torch.randn(4, 3, generator=g).requires_grad_(), torch.randn(5, generator=g).requires_grad_()]


def test_constructor_matches_torch_adamw_groups():
    ps = _params()
    ours = FusedAdamW(ps, lr=2e-4, betas=(0.8, 0.95), eps=1e-6, weight_decay=0.1)
    ref = torch.optim.AdamW(ps, lr=2e-4, betas=(0.8, 0.95), eps=1e-6, weight_decay=0.1)
    for k in ("lr", "betas", "eps", "weight_decay", "amsgrad", "maximize"):
        assert ours.param_groups[0][k] == ref.param_groups[0][k]
    assert ours.param_groups[0]["fused"] and ours._step_supports_amp_scaling  # train_step folds clipping through grad_scale
    with pytest.raises(ValueError):
        FusedAdamW(ps, lr=-1.0)
    with pytest.raises(ValueError):
        FusedAdamW(ps, betas=(1.0, 0.9))


def test_state_dict_layout_is_torch_adamw():
    ps = _params()
    ours = FusedAdamW(ps, lr=1e-3)
    for p in ps:
        ours._init_state(p)
    sd = ours.state_dict()
    assert sorted(sd["state"]) == [0, 1]
    assert sorted(sd["state"][0]) == ["exp_avg", "exp_avg_sq", "step"]
    assert "_ur_launches" not in sd["param_groups"][0]
    ref = torch.optim.AdamW([p.detach().clone().requires_grad_() for p in ps], lr=1e-3)
    ref.load_state_dict(sd)  # torch accepts it
    assert float(ref.state[ref.param_groups[0]["params"][0]]["step"]) == 0.0
    ours2 = FusedAdamW([p.detach().clone().requires_grad_() for p in ps], lr=1e-3)
    ours2.load_state_dict(ref.state_dict())
    st = ours2.state[ours2.param_groups[0]["params"][1]]
    assert st["step"].dtype == torch.float32 and st["step"].shape == ()
    # the step entries handed out are independent tensors (torch's optimizers increment each one)
    assert sd["state"][0]["step"] is not sd["state"][1]["step"]


def test_step_fails_loudly_without_a_gpu():
    ps = _params()
    for p in ps:
        p.grad = torch.ones_like(p)
    ours = FusedAdamW(ps, lr=1e-3)
    with pytest.raises(Exception) as e:  # no CPU fallback: either the library refuses to load or the dtype / device check fires
        ours.step()
    assert "GPU" in str(e.value) or "HIP" in str(e.value) or "liburhip" in str(e.value) or "hip" in str(e.value).lower()
    assert all(torch.equal(p, q) for p, q in zip(ps, _params()))  # nothing was updated


def test_pickle_drops_the_launch_cache():
    import pickle
    ps = _params()
    ours = FusedAdamW(ps, lr=1e-3)
    ours.param_groups[0]["_ur_launches"] = ("not", "picklable", (lambda: None), None)
    clone = pickle.loads(pickle.dumps(ours))
    assert "_ur_launches" not in clone.param_groups[0] and clone.param_groups[0]["lr"] == 1e-3


def test_state_dict_does_not_cut_the_shared_step_counter():
    """state_dict() used to replace the LIVE ``step`` entries with clones, so the device counter the cached
    launches increment and ``state[p]["step"]`` drifted apart (every later checkpoint froze at the first save)."""
    ps = _params()
    ours = FusedAdamW(ps, lr=1e-3)
    states = [ours._init_state(p) for p in ps]
    counter = states[0]["step"]
    states[1]["step"] = counter  # what step() does on its first call: one counter per group
    counter += 1
    sd1 = ours.state_dict()
    assert all(ours.state[p]["step"] is counter for p in ps)  # live entries untouched
    counter += 1  # the next (cached) step
    sd2 = ours.state_dict()
    assert float(sd1["state"][0]["step"]) == 1.0 and float(sd1["state"][1]["step"]) == 1.0  # snapshots stay snapshots
    assert float(sd2["state"][0]["step"]) == 2.0 and float(sd2["state"][1]["step"]) == 2.0
    assert sd2["state"][0]["exp_avg"] is ours.state[ps[0]]["exp_avg"]  # like torch: moments are handed out uncopied


def test_load_state_dict_keeps_the_addresses_a_captured_update_holds():
    """``load_state_dict`` used to drop the device (lr, weight_decay) pair and replace the moment tensors, while a
    captured ``ur_adamw_multi`` keeps the OLD addresses (stale lr / garbage moments after resume_from_checkpoint).  Now the
    existing tensors receive the loaded values in place; only genuinely new state bumps ``generation`` (-> re-capture)."""
    ps = _params()
    ours = FusedAdamW(ps, lr=1e-3)
    states = [ours._init_state(p) for p in ps]
    states[1]["step"] = states[0]["step"]
    states[0]["step"] += 3
    for st in states:
        st["exp_avg"].fill_(0.25)
        st["exp_avg_sq"].fill_(0.5)
    fake_hyper = [torch.zeros(2), (1e-3, 1e-2)]
    ours.param_groups[0]["_ur_hyper"] = fake_hyper
    sd = {"state": {k: {n: (v.clone() if torch.is_tensor(v) else v) for n, v in st.items()} for k, st in ours.state_dict()["state"].items()},
          "param_groups": ours.state_dict()["param_groups"]}
    ptrs = [(st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), st["step"].data_ptr()) for st in states]
    for st in states:  # training went on ...
        st["exp_avg"].fill_(9.0)
        st["exp_avg_sq"].fill_(9.0)
    states[0]["step"] += 5
    gen = ours.generation
    ours.load_state_dict(sd)  # ... and is rolled back to the checkpoint
    assert ours.generation == gen  # nothing a captured graph points at moved
    assert ours.param_groups[0]["_ur_hyper"] is fake_hyper and fake_hyper[1] is None  # same device pair, value to be re-sent
    for p, ptr in zip(ps, ptrs):
        st = ours.state[p]
        assert (st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), st["step"].data_ptr()) == ptr
        assert float(st["exp_avg"].flatten()[0]) == 0.25 and float(st["exp_avg_sq"].flatten()[0]) == 0.5 and float(st["step"]) == 3.0
    assert ours.state[ps[0]]["step"] is ours.state[ps[1]]["step"]
    # a fresh optimizer has no state yet: the loaded tensors are adopted and the generation moves
    fresh = FusedAdamW([p.detach().clone().requires_grad_() for p in ps], lr=1e-3)
    g0 = fresh.generation
    fresh.load_state_dict(sd)
    assert fresh.generation == g0 + 1 and float(fresh.state[fresh.param_groups[0]["params"][0]]["step"]) == 3.0
    fresh.add_param_group({"params": [torch.zeros(2, requires_grad=True)]})
    assert fresh.generation == g0 + 2


def test_load_state_dict_without_entries_for_live_state_forces_a_recapture():
    """a parameter with LIVE state loading a dict that has no (or a partial) entry for it -- a checkpoint written
    before the first step -- loses its moments inside torch's load_state_dict; the cached descriptor arrays and a captured
    ur_adamw_multi would keep the addresses of freed tensors.  The launch cache must go and ``generation`` must move."""
    ps = _params()
    ours = FusedAdamW(ps, lr=1e-3)
    for p in ps:
        ours._init_state(p)
    ours.param_groups[0]["_ur_launches"] = object()  # stands for the ctypes descriptor arrays of the last step
    empty = FusedAdamW([p.detach().clone().requires_grad_() for p in ps], lr=1e-3).state_dict()  # no state yet
    assert empty["state"] == {}
    gen = ours.generation
    ours.load_state_dict(empty)
    assert ours.generation == gen + 1 and "_ur_launches" not in ours.param_groups[0]
    # partial: only the first parameter has state in the loaded dict
    ours2 = FusedAdamW(ps, lr=1e-3)
    for p in ps:
        ours2._init_state(p)
    ours2.param_groups[0]["_ur_launches"] = object()
    sd = ours2.state_dict()
    sd = {"state": {0: sd["state"][0]}, "param_groups": sd["param_groups"]}
    gen = ours2.generation
    ours2.load_state_dict(sd)
    assert ours2.generation == gen + 1 and "_ur_launches" not in ours2.param_groups[0]
