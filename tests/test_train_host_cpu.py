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
