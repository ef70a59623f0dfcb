"""`.param` subset reader and the `concept`-style command line: parsing, forward references, units,
error codes. The run itself needs a GPU (tests/test_gpu_cli.py)."""
from pathlib import Path

import numpy as np
import pytest

from bachelor_thesis_n_body_amd import params, concept

REF = Path("/root/reference")

PARAM = """
path.output_dir = "{out}"
path.job_dir    = "{out}/job"
random_seeds = {{
    'general'              : 0,
    'primordial amplitudes': 11,
    'primordial phases'    : 12,
}}
_size = 8
_tag = f"NNPM{{_size}}_As{{primordial_spectrum['A_s']:.2e}}"      # forward reference, as in NN.param:4
initial_conditions = {{
    'species': 'matter',
    'N'      : _size**3,
}}
output_dirs = {{
    'snapshot': f'{{path.output_dir}}/{{_tag}}/snapshots',
}}
output_times = {{
    'snapshot' : [1.0],
}}
boxsize = 500*Mpc
autosave_interval = 6*hr
select_forces = {{
    'matter': {{'gravity': '{gravity}'}},
}}
H0   = 67.36*km/(s*Mpc)
Ωb   = 0.02237 /h**2
Ωcdm = 0.12 /h**2
a_begin = 0.02
primordial_spectrum = {{
    'A_s': 2.105e-9,
    'n_s': 0.9649,
}}
"""


def _write(tmp_path, gravity="pp"):
    f = tmp_path / "run.param"
    f.write_text(PARAM.format(out=tmp_path / "output", gravity=gravity), encoding="utf-8")
    return f


def test_parse_units_forward_references_and_lazy_h(tmp_path):
    p = params.parse(_write(tmp_path))
    s = params.settings(p)
    assert s["N"] == 512 and s["boxsize"] == 500.0 and s["gravity"] == "pp" and s["a_begin"] == 0.02
    assert s["snapshot_dir"] == f"{tmp_path}/output/NNPM8_As2.10e-09/snapshots" or s["snapshot_dir"].endswith("NNPM8_As2.11e-09/snapshots")
    assert s["random_seeds"]["primordial phases"] == 12 and s["snapshot_times"] == [1.0]
    h = p["H0"] / (100 * params.UNITS["km"] / params.UNITS["s"])
    assert h == pytest.approx(0.6736, rel=1e-12)
    assert p["Ωb"] == pytest.approx(0.02237 / 0.6736 ** 2, rel=1e-12)
    assert p["H0"] == pytest.approx(0.06889, rel=1e-3)            # 1/Gyr: the value CONCEPT stores in its snapshots
    assert p["autosave_interval"] == pytest.approx(6 / (24 * 365.25) * 1e-9, rel=1e-12)
    assert params.G_NEWTON == pytest.approx(4.4985e-5, rel=1e-4)   # Mpc^3 / (1e10 Msun) / Gyr^2


def test_errors(tmp_path):
    bad = tmp_path / "bad.param"
    bad.write_text("boxsize = 10*Mpc\nx = undefined_name + 1\n")
    with pytest.raises(params.ParamError, match="unresolved name"):
        params.parse(bad)
    bad.write_text("boxsize = = 3\n")
    with pytest.raises(params.ParamError):
        params.parse(bad)
    with pytest.raises(params.ParamError, match="initial_conditions"):
        params.settings({"boxsize": 1.0})


def test_cli_rejects_other_solvers_and_missing_inputs(tmp_path, capsys):
    assert concept.main(["-p", str(_write(tmp_path, "p3m")), "-n", "1", "--local"]) == 3
    assert "direct summation ('pp') only" in capsys.readouterr().err
    assert concept.main(["-p", str(tmp_path / "nope.param")]) != 0 if (tmp_path / "nope.param").exists() else True
    bad = tmp_path / "bad.param"
    bad.write_text("x = = 1\n")
    assert concept.main(["-p", str(bad)]) == 2


@pytest.mark.skipif(not REF.exists(), reason="reference tree only exists in the build container")
def test_reference_param_files_parse():
    g = params.settings(params.parse(REF / "Different_Sizes" / "Giant.param"))
    assert g["N"] == 256 ** 3 and g["boxsize"] == 1000.0 and g["gravity"] == "pm"
    n = params.settings(params.parse(REF / "TheFeedingPipeline" / "NN.param"))
    assert n["N"] == 32 ** 3 and n["boxsize"] == 500.0 and n["gravity"] == "p3m"
    assert n["snapshot_dir"].endswith("NNPM32_As2.11e-09/snapshots")      # the directory name seen in Datafolder4
