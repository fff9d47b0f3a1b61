"""How the linker takes the acquired datasets of a script (vvm_inspect_blob, no device): two to a pass where every pair's program
fits a kernel with two chains, else one program per dataset.  ProgramInfo.Flags [20:24) = passes, 0x40000 = the first pass has
two chains."""
import pytest

import dataset_cases
from renderer_amd import abi


def _plan(script):
    f = abi.load_hip_backend().inspect_blob(script.blob).Flags
    return (f >> 20) & 15, bool(f & 0x40000)


@pytest.mark.parametrize("k, passes", [(1, 1), (2, 1), (3, 2), (4, 2), (5, 3), (8, 4)])
def test_datasets_go_two_to_a_pass(k, passes):
    assert _plan(dataset_cases.many_dataset_script(k)) == (passes, k >= 2)


def test_a_transcendental_function_in_a_pair_undoes_the_pairing():
    assert _plan(dataset_cases.many_dataset_script(4, trig_in=3)) == (4, False)      # (dataset 3 shares its pass with dataset 2)
    assert _plan(dataset_cases.many_dataset_script(5, trig_in=4)) == (3, True)       # (dataset 4 has a pass of its own)


def test_without_chains_one_program_per_dataset(monkeypatch):
    monkeypatch.setenv("VVM_NO_CHAINS", "1")
    assert _plan(dataset_cases.many_dataset_script(4)) == (4, False)


def test_a_script_with_several_datasets_has_no_light_program_as_a_whole():
    """The light eight-wide kernel has one chain.  Asked for "the script's light program" (no pass named), the linker refuses a script
    that acquires two datasets -- a program for the first dataset alone would drop the second (found by the GPU fuzzer when the
    passes of such scripts got light programs of their own: the lean form of a wide record with a second dataset asks this way)."""
    be = abi.load_hip_backend()
    with pytest.raises(abi.VVMError) as e:
        be.inspect_code(dataset_cases.two_dataset_script().blob, 2)
    assert e.value.code == abi.VVM_E_UNSUPPORTED_OP
    assert not be.inspect_blob(dataset_cases.two_dataset_script().blob).Flags & 0x20000
