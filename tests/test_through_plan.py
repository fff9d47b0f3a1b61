"""Which scripts get a lean form (what they compute as a program for an eight-wide kernel, the streams they hand through left to
the copy kernel): the planning that vvm_alloc_state does, seen through vvm_inspect_blob, which needs no device."""
import struct

import pytest

import through_cases as tc
from renderer_amd import abi
from renderer_amd.assembler import FLOAT, INT32, Script

LIGHT, LEAN = 0x20000, 0x80000


def be():
    return abi.load_hip_backend()


def flags(script):
    return be().inspect_blob(script.blob).Flags


def test_the_fountain_fits_the_light_kernel_as_it_is():
    f = flags(tc.wide_record_script(0, 0))
    assert f & LIGHT and not f & LEAN


@pytest.mark.parametrize("extra_float,extra_int", [(2, 0), (6, 0), (8, 1), (14, 2), (30, 0), (40, 7)])
def test_wide_records_get_a_lean_form(extra_float, extra_int):
    f = flags(tc.wide_record_script(extra_float, extra_int))
    assert f & LEAN and not f & LIGHT
    assert (f >> 24) == extra_float + extra_int + 1      # and the lifetime, which the fountain hands through as well


def test_one_handed_through_stream_is_not_worth_a_second_launch():
    # the fountain with seven computed outputs and the lifetime handed through, plus one more stream computed with: nine registers
    s = tc.wide_record_script(1, 0, more_math=1)
    f = flags(s)
    assert not f & LEAN and not f & LIGHT


def test_more_streams_than_one_launch_of_the_copy_kernel_takes():
    f = flags(tc.wide_record_script(100, 0))
    assert f & LEAN and (f >> 24) == 101


def test_what_the_script_computes_may_need_the_general_eight_wide_kernel():
    f = flags(tc.wide_record_script(18, 1, more_math=6))      # fourteen streams computed with: beyond the light kernel
    assert f & LEAN and (f >> 24) == 14
    # ... but when the whole script still fits that kernel, a lean program that does not fit the light one gains nothing
    assert not flags(tc.wide_record_script(8, 0, more_math=6)) & LEAN
    # and what is computed must fit it: twenty streams computed with do not
    assert not flags(tc.wide_record_script(30, 0, more_math=12)) & LEAN


def test_an_output_written_twice_keeps_the_script_whole():
    s = Script()
    x = [s.input(0, FLOAT, k) for k in range(20)]
    s.acquireindex(s.const(0), 0)
    t = s.temp(); s.op("add", x[0], x[1], t)
    s.output_data(FLOAT, 0, [(t, s.output(0, FLOAT, 0))] + [(x[k], s.output(0, FLOAT, k)) for k in range(1, 20)])
    s.output_data(FLOAT, 0, [(x[3], s.output(0, FLOAT, 2))])      # output 2 a second time
    assert not flags(s.build()) & LEAN


def test_streams_that_are_not_plain_copies_stay_in_the_program():
    """No-advance inputs are not handed to the copy kernel; a 16-bit stream copied as it is (outputdata_half_from_half) is."""
    from renderer_amd.assembler import HALF
    s = Script()
    x = [s.input(0, FLOAT, k) for k in range(18)]
    h = s.input(0, HALF, 0)
    na = s.input(0, FLOAT, 18, noadvance=True)
    s.acquireindex(s.const(0), 0)
    t = s.temp(); s.op("add", x[0], x[1], t)
    s.output_data(FLOAT, 0, [(t, s.output(0, FLOAT, 0))] + [(x[k], s.output(0, FLOAT, k)) for k in range(1, 18)] + [(na, s.output(0, FLOAT, 18))])
    s.output_data(HALF, 0, [(h, s.output(0, HALF, 0))], from_half=True)
    f = flags(s.build())
    assert f & LEAN and (f >> 24) == 18      # seventeen float streams and the half one; not the no-advance input


def test_a_float_written_to_a_half_stream_is_a_conversion_not_a_copy():
    from renderer_amd.assembler import HALF
    s = Script()
    x = [s.input(0, FLOAT, k) for k in range(20)]
    s.acquireindex(s.const(0), 0)
    t = s.temp(); s.op("add", x[0], x[1], t)
    s.output_data(FLOAT, 0, [(t, s.output(0, FLOAT, 0))] + [(x[k], s.output(0, FLOAT, k)) for k in range(1, 18)])
    s.output_data(HALF, 0, [(x[18], s.output(0, HALF, 0)), (x[19], s.output(0, HALF, 1))])
    f = flags(s.build())
    assert f & LEAN and (f >> 24) == 17


def test_a_compiler_format_script_with_a_wide_record_ends_up_lean():
    """The realistic route: a script as the Niagara compiler writes it (every attribute read into a temp at the top, every attribute
    written at the bottom, also the ones nothing happens to) goes through vvm_optimize_script, whose input folding turns the
    untouched attributes into outputs that read their input directly; the blob that comes out gets the lean form."""
    from renderer_amd.original import K, OriginalScript
    o = OriginalScript()
    attrs = [o.input(0, FLOAT, i, o.temp()) for i in range(24)]
    px, py, pz, vx, vy, vz, age, life = attrs[:8]
    dt, g, drag = K(0), [K(1), K(2), K(3)], K(4)
    age2 = o.temp(); o.op("add", age, dt, age2)
    alive = o.temp(); o.op("cmplt", age2, life, alive)
    vel, pos = [], []
    for v, p, gi in zip((vx, vy, vz), (px, py, pz), g):
        t = o.temp(); o.op("mad", gi, dt, v, t)
        v2 = o.temp(); o.op("mul", t, drag, v2)
        p2 = o.temp(); o.op("mad", v2, dt, p, p2)
        vel.append(v2); pos.append(p2)
    idx = o.temp(); o.acquireindex(alive, idx, 0)
    for i, r in enumerate(pos + vel + [age2] + attrs[7:]):
        o.output(FLOAT, 0, idx, r, i)
    blob = be().optimize_script(o.done(), o.ext)
    info = be().inspect_blob(blob)
    assert info.Flags & LEAN and (info.Flags >> 24) == 17, hex(info.Flags)
    assert info.NumInputBuffers == 24
