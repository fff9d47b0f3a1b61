"""Tiles of the light eight-wide kernel that lose some of their instances: its seventh and eighth output leave a tile in one of
three ways (as they lie when nothing died; compacted within the lane when more than three quarters of the tile survive; through the
unused ends of the staging slots otherwise, 512 survivors per slot end).  Survivor counts on both sides of every boundary of that
scheme, every keep pattern of a lane's four instances, and death rates from 0.1 % to 99.9 %, against the CPU oracle."""
import numpy as np
import pytest

from renderer_amd import abi, scripts
from test_gpu_eight_wide import _check8

pytestmark = pytest.mark.gpu
f32 = np.float32
TILE = 2048


def _run(dead, instance_offset=0):
    w = scripts.fountain()
    n = dead.size
    ins = scripts.fountain_streams_np(0, n)
    ins[6] = np.where(dead, ins[7], ins[6] * f32(0.0)).astype(f32)
    got_n = _check8(w.script, [w.constants], ins, [f32] * 8, n, instance_offset=instance_offset)
    assert got_n == int((~dead).sum())


@pytest.mark.parametrize("survivors", [0, 1, 2, 3, 4, 5, 511, 512, 513, 1023, 1024, 1025, 1535, 1536, 1537, 1538, 1540, 2044, 2046, 2047])
@pytest.mark.parametrize("where", ["front", "back", "spread"])
def test_survivor_counts_around_every_boundary(survivors, where):
    """Five tiles; the second and the fourth keep exactly `survivors` instances, the others all of theirs or a random half."""
    rng = np.random.default_rng(survivors * 3 + len(where))
    dead = np.zeros(5 * TILE - 77, bool)
    dead[2 * TILE:3 * TILE] = rng.random(TILE) < 0.5
    for t in (1, 3):
        d = np.ones(TILE, bool)
        if where == "front":
            d[:survivors] = False
        elif where == "back":
            d[TILE - survivors:] = False
        else:
            d[rng.permutation(TILE)[:survivors]] = False
        dead[t * TILE:(t + 1) * TILE] = d
    _run(dead, instance_offset=3)


def test_every_keep_pattern_of_a_lane():
    """Lane l of a tile holds instances 4l .. 4l+3 and 1024 + 4l .. 1024 + 4l+3: all sixteen keep patterns in both halves, in tiles
    that keep more than three quarters (patterns thinned out with fully alive lanes) and in tiles that keep fewer."""
    for alive_lanes in (0, 3):          # of every four lanes, that many keep everything
        dead = np.zeros(4 * TILE, bool)
        for i in range(dead.size // 4):
            lane = i % 256
            if lane % 4 < alive_lanes:
                continue
            pattern = (i * 7 + i // 256) % 16
            for k in range(4):
                dead[4 * i + k] = not ((pattern >> k) & 1)
        _run(dead)


@pytest.mark.parametrize("rate", [0.001, 0.01, 0.1, 0.24, 0.26, 0.5, 0.9, 0.999])
def test_death_rates(rate):
    rng = np.random.default_rng(int(rate * 1000))
    _run(rng.random(150_001) < rate, instance_offset=1)


class _Emitter:
    """One member of a batch: the fountain on `n` instances of which a share `rate` dies, streams in device memory."""
    CAP = 20_000

    def __init__(self, be, e, n, rate):
        from renderer_amd.runtime import VectorVMState
        from vm_harness import PAD
        self.be, self.n, self.pad = be, n, PAD
        self.w = scripts.fountain(variant=e % 16)
        self.host_in = scripts.fountain_streams_np(e * 1_000_003, self.CAP)
        dead = np.random.default_rng(e).random(self.CAP) < rate
        self.host_in[6] = np.where(dead, self.host_in[7], self.host_in[6] * f32(0.0)).astype(f32)
        self.ins, self.outs = [], []
        for a in self.host_in:
            buf = np.zeros(self.CAP + PAD, f32)
            buf[:self.CAP] = a
            p = be.lib.vvm_device_alloc(buf.nbytes)
            assert p
            be._check(be.lib.vvm_device_upload(p, buf.ctypes.data, buf.nbytes), "upload")
            self.ins.append(p)
        for _ in range(8):
            p = be.lib.vvm_device_alloc((self.CAP + PAD) * 4)
            assert p
            be._check(be.lib.vvm_device_memset(p, 0xCD, (self.CAP + PAD) * 4), "memset")
            self.outs.append(p)
        self.state = VectorVMState(self.w.script.blob, be)
        self.frame = self.state.frame([abi.DataSet(self.ins, self.outs, 0)], n, [self.w.constants])

    def check(self, tag):
        from vm_harness import run_oracle
        ref, ref_n, _ = run_oracle(self.w.script, [self.w.constants], [a[:max(self.n, 1)] for a in self.host_in], [f32] * 8, self.n)
        assert self.frame.survivors(0) == ref_n, tag
        for k in range(8):
            h = np.empty(self.CAP + self.pad, np.uint32)
            self.be._check(self.be.lib.vvm_device_download(h.ctypes.data, self.outs[k], h.nbytes), "download")
            assert np.array_equal(h[:ref_n], ref[k][:ref_n].view(np.uint32)), (tag, k)
            assert (h[ref_n:] == 0xCDCDCDCD).all(), (tag, k)

    def free(self):
        self.state.close()
        for p in self.ins + self.outs:
            self.be.lib.vvm_device_free(p)


def test_members_of_a_batch_that_lose_instances_at_every_rate():
    """Members of one launch of the light kernel: a workgroup finishes one member's last tile (its two riding outputs by any of
    the three ways) while it runs the next member's first tile, whose death rate is another one."""
    be = abi.load_hip_backend()
    rates = [0.0, 0.03, 0.15, 0.3, 0.6, 0.97, 0.01, 0.5]
    sizes = [2048, 4096, 6000, 10_000, 2049, 12_289, 17_000, 3000, 8192, 5]
    ems = [_Emitter(be, e, sizes[e % len(sizes)], rates[e % len(rates)]) for e in range(40)]
    handle = None
    try:
        opts = [abi.ExecOptions(0, 0, 0, None, abi.VVM_EXEC_FORCE_EIGHT_WIDE) for _ in ems]
        arr, oarr = be.pack_batch([em.frame for em in ems], opts)
        be.exec_batch_packed(arr, oarr)
        for i, em in enumerate(ems):
            em.check("plain batch, emitter %d" % i)
            assert em.state.last_exec_width() == 8 and em.state.info().Flags & 0x20000
        handle = be.batch_create(arr, oarr)
        for i, em in enumerate(ems):
            em.n = min(em.CAP, em.n + 1000 + 13 * i)
            arr[i].NumInstances = em.n
        be.batch_exec(handle, abi.VVM_BATCH_CHANGED_INSTANCES)
        for i, em in enumerate(ems):
            em.check("handle, emitter %d" % i)
    finally:
        if handle:
            be.batch_destroy(handle)
        for em in ems:
            em.free()
