"""Host logic of the epoch-level wrappers (pointcloududa_amd/_epoch.py): adopting the caller's torch.optim objects,
the torch-format optimiser state, the batch feeder.  CPU only: nothing here launches a kernel."""
import numpy as np
import pytest
import torch


def _disc():
    from pointcloududa_amd.networks import UncertaintyDiscriminator
    torch.manual_seed(0)
    return UncertaintyDiscriminator(in_channel=4)


def test_adopt_reads_hyperparameters_and_state_of_torch_optimisers():
    from pointcloududa_amd import _epoch as E
    from pointcloududa_amd.optim import FusedAdam, FusedSGD
    d = _disc()
    sgd = torch.optim.SGD(d.parameters(), lr=3e-5, momentum=0.9, weight_decay=1e-3)
    for p in d.parameters():                       # one stock step so that the optimiser holds momentum buffers
        p.grad = torch.full_like(p, 0.5)
    sgd.step()
    f = E.adopt_optimizer(sgd, d)
    assert isinstance(f, FusedSGD) and (f.lr, f.momentum, f.wd) == (3e-5, 0.9, 1e-3) and f.steps == 1
    for (o, n, shp), p in zip(f._slices(), d.parameters()):
        assert torch.equal(f.buf[o:o + n].view(shp), sgd.state[p]["momentum_buffer"])
        assert p.data_ptr() == f.p[o:o + n].data_ptr()          # the caller's Parameter objects now view the flat buffer
    sgd.param_groups[0]["lr"] *= 0.2                              # train_mscmrseg.py:585-589
    E.adopt_hyperparameters(f, sgd)
    assert abs(f.lr - 6e-6) < 1e-18
    d2 = _disc()
    adam = torch.optim.Adam(d2.parameters(), lr=1e-3, betas=(0.9, 0.99))
    f2 = E.adopt_optimizer(adam, d2)
    assert isinstance(f2, FusedAdam) and f2.betas == (0.9, 0.99) and f2.eps == 1e-8 and int(f2.step_t) == 0
    with pytest.raises(ValueError):
        E.adopt_optimizer(torch.optim.SGD(_disc().parameters(), lr=1.0), d2)       # built from another module
    with pytest.raises(TypeError):
        E.adopt_optimizer(torch.optim.RMSprop(d2.parameters(), lr=1.0), d2)


def test_export_state_round_trips_through_the_callers_optimiser():
    from pointcloududa_amd import _epoch as E
    d = _disc()
    sgd = torch.optim.SGD(d.parameters(), lr=2.5e-5, momentum=0.99, weight_decay=5e-4)
    f = E.adopt_optimizer(sgd, d)
    f.buf.copy_(torch.randn(f.buf.shape, generator=torch.Generator().manual_seed(3))); f.steps = 4
    E.export_state(f, sgd)
    for (o, n, shp), p in zip(f._slices(), d.parameters()):
        assert torch.equal(sgd.state[p]["momentum_buffer"], f.buf[o:o + n].view(shp))
    assert sgd.param_groups[0]["lr"] == 2.5e-5


def test_torch_format_state_omits_parameters_the_reference_never_updates():
    """torch.optim holds no state for a parameter whose .grad is None (encoder.conv1_1; the point head without a loss on
    it): the exported state dicts must not either, and stock optimisers must load them."""
    from oracle import nets as ON
    from pointcloududa_amd.networks import Segmentation_model_Point
    from pointcloududa_amd.optim import FusedAdam, FusedSGD
    kw = dict(filters=4, in_channels=1, n_class=4, pointnet=True, fc_inch=9)
    m = Segmentation_model_Point(**kw)
    names = [k for k, _ in m.named_parameters()]
    o = FusedSGD(m, lr=1e-3, momentum=0.95, weight_decay=5e-4, skip_prefixes=("encoder.conv1_1.", "pointNet."))
    o.buf.fill_(0.25); o.steps = 2
    sd = o.torch_state_dict()
    skipped = [i for i, k in enumerate(names) if k.startswith(("encoder.conv1_1.", "pointNet."))]
    assert skipped and all(i not in sd["state"] for i in skipped) and len(sd["state"]) == len(names) - len(skipped)
    ref = torch.optim.SGD([torch.nn.Parameter(p.detach().clone()) for p in m.parameters()], lr=1.0, momentum=0.5)
    ref.load_state_dict(sd)
    assert len(ref.state_dict()["state"]) == len(names) - len(skipped)
    full = {i: {"momentum_buffer": torch.ones(shp)} for i, (_, _, shp) in enumerate(o._slices())}
    o.load_torch_state_dict({"state": full, "param_groups": sd["param_groups"]})
    for i, (off, n, _) in enumerate(o._slices()):          # skipped ranges stay zero whatever the file holds
        assert float(o.buf[off:off + n].abs().max()) == (0.0 if i in skipped else 1.0)
    m2 = Segmentation_model_Point(**kw)
    a = FusedAdam(m2, lr=1e-3)
    a.step_t.fill_(3)
    for i, (off, n, _) in enumerate(a._slices()):
        if not names[i].startswith("encoder.conv1_1."):
            a.v[off:off + n].fill_(1e-4); a.m[off:off + n].fill_(1e-2)
    sda = a.torch_state_dict()
    unused = [i for i, k in enumerate(names) if k.startswith("encoder.conv1_1.")]
    assert unused and all(i not in sda["state"] for i in unused) and len(sda["state"]) == len(names) - len(unused)
    torch.optim.Adam([torch.nn.Parameter(p.detach().clone()) for p in m2.parameters()], lr=1.0).load_state_dict(sda)


def test_device_batches_feed_numpy_generators_in_order():
    from pointcloududa_amd._epoch import DeviceBatches
    rng = np.random.default_rng(0)
    items = [(rng.random((2, 3, 8, 8)).astype(np.float32), (rng.random((2, 4, 8, 8)) > 0.5), rng.random((2, 300, 3)))
             for _ in range(5)]
    got = list(DeviceBatches(iter(items), torch.device("cpu")))
    assert len(got) == 5
    for (x, y, z), (gx, gy, gz) in zip(items, got):
        assert gx.dtype == torch.float32 and gy.dtype == torch.uint8 and gz.dtype == torch.float32
        assert np.array_equal(gx.numpy(), x) and np.array_equal(gy.numpy(), y.astype(np.uint8))
        assert np.allclose(gz.numpy(), z.astype(np.float32))


class _RefProtocolGenerator:
    """The iterator protocol of the reference's data generators (data_generator_mscmrseg.py:270-319,
    data_generator_mmwhs.py:213-274): ``__iter__`` resets the epoch count, ``__next__`` resets it AS IT RAISES
    StopIteration -- so one more ``next()`` starts a new epoch -- and ``_index`` walks on modulo the data set."""

    def __init__(self, n_batches, length=1000):
        self.n, self.len = n_batches, length
        self._totalcount = self._index = self.calls = 0

    def __iter__(self):
        self._totalcount = 0
        return self

    def __next__(self):
        self.calls += 1
        if self._totalcount >= self.n:
            self._totalcount = 0
            raise StopIteration
        self._totalcount += 1
        i, self._index = self._index, (self._index + 1) % self.len
        return (np.full((2, 1, 4, 4), i, np.float32), np.zeros((2, 4, 4, 4), np.uint8), np.zeros((2, 300, 3), np.float32))


def test_device_batches_stop_at_the_first_stopiteration_of_a_reference_style_generator():
    from pointcloududa_amd._epoch import DeviceBatches
    g = _RefProtocolGenerator(5)
    got = [int(x[0, 0, 0, 0]) for x, _, _ in DeviceBatches(g, torch.device("cpu"))]
    assert got == [0, 1, 2, 3, 4] and g.calls == 6             # five batches + the ONE call that raised
    got = [int(x[0, 0, 0, 0]) for x, _, _ in DeviceBatches(g, torch.device("cpu"))]
    assert got == [5, 6, 7, 8, 9] and g.calls == 12            # the next epoch continues where zip would have left it


def test_paired_read_ahead_advances_the_generators_exactly_as_zip_does():
    # train_epoch pairs the iterators as the reference's zip(trainA_iterator, trainB_iterator) does
    from pointcloududa_amd._epoch import DeviceBatches
    for na, nb in ((5, 7), (7, 5), (4, 4)):
        a, b = _RefProtocolGenerator(na), _RefProtocolGenerator(nb)
        ra, rb = _RefProtocolGenerator(na), _RefProtocolGenerator(nb)
        for _epoch in range(3):
            want = [(int(x[0][0, 0, 0, 0]), int(y[0][0, 0, 0, 0])) for x, y in zip(ra, rb)]
            got = [(int(t[0][0, 0, 0, 0]), int(t[3][0, 0, 0, 0]))
                   for t in DeviceBatches((tuple(x) + (y[0], None, y[2]) for x, y in zip(a, b)), torch.device("cpu"))]
            assert got == want and len(got) == min(na, nb)
            assert (a._index, a._totalcount, a.calls) == (ra._index, ra._totalcount, ra.calls)
            assert (b._index, b._totalcount, b.calls) == (rb._index, rb._totalcount, rb.calls)


def test_train_epoch_needs_args_like_the_reference_module_global():
    from pointcloududa_amd import train_mmwhs, train_mscmrseg
    for mod in (train_mscmrseg, train_mmwhs):
        mod.args = None
        with pytest.raises(ValueError):
            mod.train_epoch(None, None, None)
