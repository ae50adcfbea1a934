"""Host-side helpers of the training path that need no GPU."""
import pytest
import torch

from uni_renderer_amd import train_step as TS


class _Net(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.lin = torch.nn.Linear(8, 8)
        self.c1 = torch.nn.Conv2d(8, 8, 1)
        self.c3 = torch.nn.Conv2d(8, 8, 3, padding=1)
        self.norm = torch.nn.LayerNorm(8)


def test_parameter_list_is_cached_per_network_tuple():
    nets = (_Net(), _Net(), _Net())
    ps = TS._parameters(nets)
    assert len(ps) == sum(1 for n in nets for _ in n.parameters())
    assert TS._parameters(nets) is ps                      # same tuple of networks: the cached list
    other = (_Net(), nets[1], nets[2])
    assert TS._parameters(other) is not ps and TS._parameters(other)[0] is other[0].lin.weight


def test_batched_casts_cover_linear_and_1x1_weights_and_fail_loudly_on_cpu():
    net = _Net()
    with torch.no_grad():                                  # inference: nothing is cast up front
        with TS._batched_casts(net, torch.bfloat16):
            assert TS._cast == {}
    with pytest.raises(RuntimeError):                      # training on CPU tensors: no CPU fallback
        with TS._batched_casts(net, torch.bfloat16):
            pass
    ws = TS._castable[id(net)][1]
    assert {id(w) for w in ws} == {id(net.lin.weight), id(net.c1.weight)}  # Linear + 1x1 conv; not the 3x3, not the norm
    assert TS._cast == {}
    w = TS._wc(net.lin.weight, torch.bfloat16)             # outside a batched cast: the plain differentiable cast
    assert w.dtype == torch.bfloat16 and w.requires_grad


def test_derived_weight_cache_never_serves_a_dead_buffer():
    """the W^T / rotated-weight caches are keyed by raw address; an entry must die with the buffer it was derived
    from (a later tensor at the same address with the same shape would otherwise pick up a stale transpose)."""
    import gc

    import torch

    from uni_renderer_amd.backward import _DerivedWeights

    cache = _DerivedWeights()
    owner = torch.zeros(4, 4)
    key = (owner.data_ptr(), (4, 4))
    cache.put(key, owner, torch.ones(4, 4))
    assert torch.equal(cache.lookup(key), torch.ones(4, 4))
    del owner
    gc.collect()
    assert cache.lookup(key) is None and key not in cache  # the dead entry is a miss and is dropped
    o2 = torch.zeros(2, 2)
    cache.put((1, (2, 2)), o2, torch.ones(2, 2))
    o3 = torch.zeros(3)
    cache.put((2, (3,)), o3, torch.ones(3))
    del o3
    gc.collect()
    cache.sweep()
    assert list(cache) == [(1, (2, 2))]


def test_norm_sums_defer_each_parameter_once_and_never_under_anomaly_detection():
    """a norm layer (or tied gamma / beta) used twice before a flush must not hand out two uninitialised results
    that autograd adds at once; anomaly detection inspects node outputs immediately."""
    import torch

    from uni_renderer_amd.backward import NormSums

    ns = NormSums()
    g1, g2 = torch.ones(8), torch.ones(8)
    assert ns.fresh(g1) and ns.fresh(g2)
    assert not ns.fresh(g1)  # the repeat takes the immediate path
    ns.reset()
    assert ns.fresh(g1)
    ns.reset()
    with torch.autograd.detect_anomaly(check_nan=False):
        assert not ns.fresh(g1)


def test_grad_sink_first_writer_stores_later_writers_add():
    """Round 6: gradients living in communication buckets are WRITTEN there by the kernels that produce them
    (backward.GradSink) instead of being added into zeroed buckets by autograd.  Host logic: the first ``take`` of a parameter
    since ``begin`` hands out the destination, later ones (second pass of the cycle branch, accumulation micro-steps) get None
    = the ordinary add path; anomaly mode and ``end()`` switch the sink off."""
    import torch
    from uni_renderer_amd import backward as B

    store = {1: torch.zeros(3), 2: torch.zeros(2)}
    sink = B.GradSink()
    assert sink.take(1) is None  # no provider installed
    sink.begin(lambda pid: store.get(pid))
    assert sink.take(1) is store[1] and sink.take(1) is None and sink.take(3) is None and sink.take(2) is store[2]
    sink.begin(lambda pid: store.get(pid))  # next optimisation step
    with torch.autograd.detect_anomaly(check_nan=False):
        assert sink.take(1) is None
    assert sink.take(1) is store[1]
    sink.end()
    assert sink.take(2) is None
    # cast_many(outs=...) validates its destinations before touching the device
    import pytest
    with pytest.raises(ValueError):
        B.cast_many([torch.zeros(4, dtype=torch.bfloat16)], torch.float32, outs=[torch.zeros(3)])
