"""CPU oracle for the PointCloudUDA adversarial train-step hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``pointcloududa_amd/`` may import this
package; only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` do, and only as the checker.

The oracle restates, in plain PyTorch-CPU / numpy and in a functional style over
parameter dictionaries (same ``state_dict`` keys as the reference modules), the
arithmetic of

* ``src/networks/unet.py``        -> :mod:`oracle.nets` (``seg_forward``)
* ``src/networks/GAN.py``         -> :mod:`oracle.nets` (``disc_forward``)
* ``src/networks/PointNetCls.py`` -> :mod:`oracle.nets` (``pointnet_cls_forward``)
* ``src/utils/loss.py``           -> :mod:`oracle.losses`
* ``src/utils/metric.py:5-36``, ``src/utils/utils.py:32-40`` -> :mod:`oracle.metrics`
* ``src/utils/npy2point.py:7-18,101-125`` -> :mod:`oracle.sampler`
* ``src/train_mscmrseg.py:183-330`` / ``src/train_mmwhs.py:187-360`` -> :mod:`oracle.step`

Parity pinning: ``oracle/make_golden.py`` imports the real reference modules
from ``/root/reference/src`` (possible only in the build container), loads the
same numpy-seeded weights into both, checks that the restatement agrees with
the reference (forward, backward and one full 5-phase step) and writes the
``tests/golden/*.npz`` fixtures that travel to the GPU box.  The one piece that
is *parity unpinned* is the marching-cubes vertex extraction (PyMCubes is not
vendored and not installed): see :mod:`oracle.sampler`.
"""
