"""CPU: guards on the generated gfx950 ISA of csrc/kernels.hip (hipcc cross-compiles without a GPU).

1. The one hazard the compiler cannot see: the diagonal block's v_fmac_f64_dpp / v_mov_b64_dpp live in asm statements with
   hand-placed s_nop -- a DPP operand read needs two wait states behind the VALU write of its source (CDNA3 ISA 4.5).  A compiler
   upgrade or an edit that reorders the asm would ship silently; tools/dpp_hazard_check.py scans every function.
2. Register spills of the default task-queue instantiations: dag_kernel<double> / dag_kernel<float> must keep their
   accumulators in registers (0 spilled VGPRs, no scratch)."""
import importlib.util
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


@pytest.fixture(scope="module")
def isa():
    if not (os.path.exists(HIPCC) or shutil.which("hipcc")):
        pytest.skip("hipcc not available")
    subprocess.check_call(["make", "-C", ROOT, "build/kernels.s"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    with open(os.path.join(ROOT, "build", "kernels.s")) as f:
        return f.read()


def test_no_dpp_read_within_two_wait_states_of_its_source(isa):
    spec = importlib.util.spec_from_file_location("dpp_hazard_check", os.path.join(ROOT, "tools", "dpp_hazard_check.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    n_dpp, n_bad = mod.check(isa.split("\n"))
    assert n_dpp > 1000, "the diagonal block's DPP eliminations were not found in the ISA: the scan would be vacuous"
    assert n_bad == 0


def _kernel_notes(isa):
    """name -> {key: int} from the amdhsa metadata (one YAML block per kernel)."""
    out = {}
    meta = isa[isa.index("amdhsa.kernels:"):]
    for block in re.split(r"\n  - ", meta)[1:]:
        name = re.search(r"\.name:\s+(\S+)", block)
        if not name:
            continue
        out[name.group(1)] = {k: int(v) for k, v in re.findall(r"\.(\w+_count|private_segment_fixed_size):\s+(\d+)", block)}
    return out


def test_default_task_queue_kernels_do_not_spill(isa):
    notes = _kernel_notes(isa)
    for tname in ("d", "f"):
        sym = f"_ZN5hbegp10dag_kernelI{tname}EEvNS_9DagLaunchE"
        assert sym in notes, sorted(k for k in notes if "dag_kernel" in k)
        assert notes[sym]["vgpr_spill_count"] == 0, (sym, notes[sym])
        assert notes[sym]["vgpr_count"] <= 256
