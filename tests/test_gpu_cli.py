"""End-to-end run of the `concept`-style command line on the GPU: parameter file in, CONCEPT-layout snapshot +
power-spectrum table out."""
import numpy as np
import pytest

from bachelor_thesis_n_body_amd import concept, snapshot as sn
from test_params_cli import _write

pytestmark = pytest.mark.gpu


def test_cli_generates_ics_runs_and_writes_outputs(tmp_path):
    k = np.logspace(-3, 1, 200)
    np.savetxt(tmp_path / "pk.txt", np.column_stack([k, 2.0e4 * (k / 0.02) / (1 + (k / 0.02) ** 2) ** 1.6]))
    f = _write(tmp_path)
    rc = concept.main(["-p", str(f), "-n", "1", "--local", "--pk-table", str(tmp_path / "pk.txt"), "--steps", "3", "--dt", "0.01",
                       "-c", "output_dirs['powerspec'] = output_dirs['snapshot'].replace('snapshots', 'powerspec')"])
    assert rc == 0
    snaps = list((tmp_path / "output").rglob("snapshot_a=1.00.hdf5"))
    assert len(snaps) == 1
    pos, mom = sn.read_positions(str(snaps[0])), sn.read_momenta(str(snaps[0]))
    assert pos.shape == (512, 3) and mom.shape == (512, 3) and sn.read_boxsize(str(snaps[0])) == 500.0
    assert pos.min() >= 0 and pos.max() < 500.0 and np.all(np.isfinite(mom)) and np.abs(mom).max() > 0
    ps = list((tmp_path / "output").rglob("powerspec_a=1.00"))
    assert len(ps) == 1
    tab = np.loadtxt(ps[0])
    assert tab.shape[1] == 3 and tab[:, 1].sum() in (8 ** 3 - 1, 8 ** 3 - 2)   # np.histogram may drop the corner mode (edge rounding), as in the reference
    # restart from the written snapshot (initial_conditions by file), tree solver, fp32
    rc = concept.main(["-p", str(f), "--ic", str(snaps[0]), "--steps", "2", "--dtype", "f32", "--solver", "tree"])
    assert rc == 0
