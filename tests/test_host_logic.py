"""Host-side logic that needs no GPU: the KITTI split tables and the flat gradient buffer's accumulation mode."""
import torch


def test_kitti_test_split_is_the_reference_drive_table():
    """reference: gans/datasets/kitti.py:246-252 -- city / road / residential drives that are not train/val drives,
    category by category.  Counts and a few landmarks of the reference's table (28 + 12 + 21 drives, of which the 10
    odometry drives are removed: 2011_09_30_drive_0016 (road), 2011_10_03_drive_0042 (road), the residential ones)."""
    from gans.datasets.kitti import _ODOMETRY_TO_RAW, test_drives
    names = test_drives()
    assert len(names) == len(set(names)) == 51
    assert names[0] == "2011_09_26_drive_0001_sync" and names[27] == "2011_09_29_drive_0071_sync"   # city block
    assert names[28] == "2011_09_26_drive_0015_sync"                                                 # road block
    assert not {v[0] for v in _ODOMETRY_TO_RAW.values()} & set(names)
    assert all(n.endswith("_sync") and "_drive_" in n for n in names)
    assert not any("2011_09_28_drive_0053" in n for n in names)     # a `person` drive: not part of the test split


def test_flat_gradient_buffer_accumulates_chunks():
    """reference: trainer.py:255-257,296 -- the chunk losses are divided by the number of chunks and their gradients
    summed; FlatGradSync.collect(accumulate, scale) does that on the flat buffer, incl. parameters a chunk left
    without gradient."""
    from gans import parallel
    m = torch.nn.Sequential(torch.nn.Linear(3, 4), torch.nn.Linear(4, 2))
    sync = parallel.FlatGradSync(m)
    xs = [torch.randn(5, 3) for _ in range(3)]
    want = None
    for x in xs:
        g = torch.autograd.grad(m(x).square().sum() / 3, list(m.parameters()))
        flat = torch.cat([t.reshape(-1) for t in g])
        want = flat if want is None else want + flat
    for j, x in enumerate(xs):
        sync.begin()
        m(x).square().sum().backward()
        if j == 1:
            m[1].bias.grad = None            # a parameter without gradient in this chunk keeps its running sum
            want = want - torch.cat([torch.zeros(3 * 4 + 4 + 4 * 2), torch.autograd.grad(
                m(x).square().sum() / 3, [m[1].bias])[0]])
        sync.collect(accumulate=j > 0, scale=1.0 / 3)
    assert torch.allclose(sync.flat, want, rtol=1e-5, atol=1e-6)
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(sync.params, sync._views()))
    # a single chunk: plain copy, missing gradients zeroed
    sync.begin()
    m(xs[0]).square().sum().backward()
    m[0].weight.grad = None
    sync.collect()
    assert float(sync.flat[:12].abs().max()) == 0.0 and float(sync.flat[12:].abs().max()) > 0


def test_dilation_and_init_weights_operator_api():
    """ops.Dilation (reference common.py:256-271: a grouped transposed conv with a one-hot-plus-`value` kernel, stride
    dilation + 1, padding 1) built as a strided placement + window sums, against that definition; ops.init_weights
    (common.py:274-292)."""
    import torch
    import torch.nn as nn
    import torch.nn.functional as F
    from gans.models import ops
    torch.manual_seed(0)
    x = torch.randn(2, 3, 5, 7)
    for d, v in ((1, 0), (2, 0), (1, 0.5), (3, -0.25)):
        m = ops.Dilation(d, v)
        k = F.pad(torch.ones(1, 1, 1, 1), (d,) * 4, value=v).repeat(3, 1, 1, 1)
        want = F.conv_transpose2d(x, k, stride=d + 1, padding=1, groups=3)
        got = m(x)
        assert got.shape == want.shape and float((got - want).abs().max()) < 1e-6
        assert tuple(m.state_dict()["kernel"].shape) == (1, 1, 2 * d + 1, 2 * d + 1)
    net = nn.Sequential(nn.Conv2d(4, 8, 3), nn.Linear(8, 8))
    ops.init_weights(net, "ortho", gain=2.0)
    w = net[1].weight
    assert float((w @ w.t() - 4.0 * torch.eye(8)).abs().max()) < 1e-4 and float(net[0].bias.abs().max()) == 0.0
    ops.init_weights(net, "N02")
    assert 0.005 < float(net[0].weight.std()) < 0.04
    ops.init_weights(net, "glorot")
    bound = (6.0 / (8 + 8)) ** 0.5
    assert float(net[1].weight.abs().max()) <= bound + 1e-6
