"""N > 1 path on CPU: world_size-2 (and 3) gloo processes run the frame-sharding + gatherv logic of
picsong_dist with the oracle standing in for the GPU encoder, and the assembled video codestream
must equal the single-process result frame by frame (header only on frame 0, _SIZE in frame order)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
W, H, WL, NF = 128, 64, 1, 5


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, outdir, pipelined=False):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(ROOT, "cuda-image-and-video-codec_amd", "python"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as orc
    import picsong_dist as pd
    lut = orc.lut_for(False, WL)
    encoded = []

    fetched = []

    def encode_now(f, it):
        s = orc.encode_frame(orc.gen_frame(W, H, f), WL, False, 1.0, lut, it, NF)
        return torch.from_numpy(s.view(np.int16).copy())

    def encode_fn(f, it):
        encoded.append(f)
        if not pipelined:
            return encode_now(f, it)

        def fetch():                                   # "wait for the frame" of the pipelined form
            fetched.append((f, len(encoded)))
            return encode_now(f, it)
        return fetch

    chunks = {}
    sizes = pd.encode_video_distributed(NF, encode_fn, rank, world, torch.device("cpu"),
                                        on_frame=lambda f, s: chunks.__setitem__(f, s.numpy().view(np.uint16).copy()),
                                        pipelined=pipelined)
    if pipelined and len(encoded) > 1:
        # the exchange (and with it the wait) of a frame ran only after the NEXT frame was launched
        mine = pd.shard_frames(NF, rank, world)
        for (f, launched) in fetched[:-1]:
            assert launched >= mine.index(f) + 2, (f, launched)
    assert encoded == pd.shard_frames(NF, rank, world)
    if rank == 0:
        assert sorted(chunks) == list(range(NF))
        np.save(os.path.join(outdir, "video.npy"), np.concatenate([chunks[f] for f in range(NF)]))
        with open(os.path.join(outdir, "video_SIZE"), "w") as fh:
            fh.write(pd.size_sidecar(sizes))
    else:
        assert sizes is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,pipelined", [(2, False), (3, False), (2, True), (3, True)])
def test_frame_sharding_gather_matches_single_process(oracle, tmp_path, world, pipelined):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), pipelined), nprocs=world, join=True)
    lut = oracle.lut_for(False, WL)
    ref = [oracle.encode_frame(oracle.gen_frame(W, H, f), WL, False, 1.0, lut, 0 if f == 0 else 1, NF)
           for f in range(NF)]
    got = np.load(os.path.join(tmp_path, "video.npy"))
    assert np.array_equal(got, np.concatenate(ref))
    sizes = open(os.path.join(tmp_path, "video_SIZE")).read()
    assert sizes == ",".join(str(r.size) for r in ref) and not sizes.endswith("\n")
    # only frame 0 carries the populated header; the others have 0xFFFF there
    assert (ref[1][:9] == 0xFFFF).all() and not (ref[0][:9] == 0xFFFF).all()
    # every frame decodes back to its input
    for f in (0, NF - 1):
        assert np.array_equal(oracle.decode_frame(ref[f], W, H, WL, False, 1.0, lut), oracle.gen_frame(W, H, f))


def _step_worker(rank, world, port, outdir, chunk=64):
    """One bucketed exchange (picsong_dist.gather_step): every rank's step of streams, ragged lengths and an empty one."""
    sys.path.insert(0, os.path.join(ROOT, "cuda-image-and-video-codec_amd", "python"))
    import picsong_dist as pd
    pd.GATHER_CHUNK_FRAMES = chunk                           # (2: the step's operations go out in three grouped batches)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    rng = np.random.default_rng(100 + rank)
    lens = [int(x) for x in rng.integers(0, 4000, 6)]
    lens[rank % 6] = 0                                       # a frame without payload
    streams = [torch.from_numpy(rng.integers(-32768, 32767, n).astype(np.int16)) for n in lens]
    payload = torch.cat(streams) if sum(lens) else torch.empty(0, dtype=torch.int16)
    for use_bufs in (False, True):
        bufs = [torch.empty(6 * 4000, dtype=torch.int16) for _ in range(world - 1)] if (use_bufs and rank == 0) else None
        got = pd.gather_step(streams, rank, world, dev, recv_bufs=bufs)
        if rank == 0:
            assert len(got) == world and all(len(g) == 6 for g in got)
            np.save(os.path.join(outdir, f"got_{int(use_bufs)}.npy"),
                    np.concatenate([v.numpy() for g in got for v in g]), allow_pickle=False)
            np.save(os.path.join(outdir, f"lens_{int(use_bufs)}.npy"), np.array([[v.numel() for v in g] for g in got]))
        else:
            assert got is None
    np.save(os.path.join(outdir, f"sent_{rank}.npy"), payload.numpy())
    np.save(os.path.join(outdir, f"sentlens_{rank}.npy"), np.array(lens))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,chunk", [(2, 64), (3, 64), (3, 2)])
def test_gather_step_moves_a_whole_step_in_one_exchange(tmp_path, world, chunk):
    """gather_step: one all-gather of [world, n] lengths + grouped batches of per-frame messages (`chunk` frames a
    batch); rank 0 gets every rank's per-frame views in rank and frame order (with and without preallocated buffers)."""
    port = _free_port()
    mp.spawn(_step_worker, args=(world, port, str(tmp_path), chunk), nprocs=world, join=True)
    sent = np.concatenate([np.load(tmp_path / f"sent_{r}.npy") for r in range(world)])
    lens = np.stack([np.load(tmp_path / f"sentlens_{r}.npy") for r in range(world)])
    for u in (0, 1):
        assert np.array_equal(np.load(tmp_path / f"got_{u}.npy"), sent)
        assert np.array_equal(np.load(tmp_path / f"lens_{u}.npy"), lens)


def _rotate_worker(rank, world, port, outdir, chunk=64):
    """picsong_dist.gather_step(rotate=True): frame f of every rank's step lands on rank f mod world."""
    sys.path.insert(0, os.path.join(ROOT, "cuda-image-and-video-codec_amd", "python"))
    import picsong_dist as pd
    pd.GATHER_CHUNK_FRAMES = chunk
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cpu")
    n = 7                                                    # not a multiple of the world size
    rng = np.random.default_rng(200 + rank)
    lens = [int(x) for x in rng.integers(1, 3000, n)]
    lens[(rank + 2) % n] = 0                                 # a frame without payload
    streams = [torch.from_numpy(rng.integers(-32768, 32767, k).astype(np.int16)) for k in lens]
    for f, t in enumerate(streams):
        np.save(os.path.join(outdir, f"sent_{rank}_{f}.npy"), t.numpy())
    for use_bufs in (False, True):
        per_peer = (n + world - 1) // world * 3008
        bufs = [torch.empty(per_peer, dtype=torch.int16) for _ in range(world - 1)] if use_bufs else None
        got = pd.gather_step(streams, rank, world, dev, recv_bufs=bufs, rotate=True)
        assert len(got) == world and all(len(g) == n for g in got)
        for r in range(world):
            for f in range(n):
                if f % world == rank:
                    np.save(os.path.join(outdir, f"got{int(use_bufs)}_{rank}_{r}_{f}.npy"), got[r][f].numpy().copy())
                else:
                    assert got[r][f] is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,chunk", [(2, 64), (3, 64), (3, 3), (2, 1)])
def test_gather_step_with_the_writer_role_rotating(tmp_path, world, chunk):
    """Every (rank, frame) stream arrives, whole, at rank frame mod world and nowhere else (both buffer forms; the
    step's operations in one grouped batch or cut into several at the same frame indices on every rank)."""
    port = _free_port()
    mp.spawn(_rotate_worker, args=(world, port, str(tmp_path), chunk), nprocs=world, join=True)
    for u in (0, 1):
        for r in range(world):
            for f in range(7):
                want = np.load(tmp_path / f"sent_{r}_{f}.npy")
                got = np.load(tmp_path / f"got{u}_{f % world}_{r}_{f}.npy")
                assert np.array_equal(got, want)


def test_deferred_exchange_order():
    sys.path.insert(0, os.path.join(ROOT, "cuda-image-and-video-codec_amd", "python"))
    import picsong_dist as pd
    dx, log = pd.DeferredExchange(), []
    assert dx.submit(lambda: log.append("a") or "A") is None and log == []
    assert dx.submit(lambda: log.append("b") or "B") == "A" and log == ["a"]
    assert dx.flush() == "B" and log == ["a", "b"] and dx.flush() is None


def test_shard_helpers():
    sys.path.insert(0, os.path.join(ROOT, "cuda-image-and-video-codec_amd", "python"))
    import picsong_dist as pd
    assert pd.shard_frames(10, 1, 4) == [1, 5, 9]
    assert pd.shard_frames(3, 3, 4) == []
    assert [pd.owner_of(f, 8) for f in (0, 7, 8, 255)] == [0, 7, 0, 7]
    assert pd.size_sidecar([12, 3456, 7]) == "12,3456,7"


# ---- intra-frame sharding (BASELINE config 5 in miniature) -------------------------------------
SW, SH, SWL = 256, 192, 2          # 4 x 3 = 12 codeblocks


def _stripe_worker(rank, world, port, outdir):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(ROOT, "cuda-image-and-video-codec_amd", "python"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import oracle_lib as orc
    import picsong_dist as pd
    lut = orc.lut_for(False, SWL)
    img = orc.pad_frame(orc.gen_frame(SW, SH, 9))
    coef = orc.dwt_forward(orc.level_shift_fwd(img, False), SWL)[:SW * SH].reshape(SH, SW)
    staging, sizes = orc.bpc_encode(coef, SWL, lut)          # every rank has the whole frame's DWT

    def encode_stripe(b, n):                                  # oracle stand-in for encode_frame_stripe
        mini = orc.bitstream_pack(staging[b * 4096:(b + n) * 4096], sizes[b:b + n], None)
        return torch.from_numpy(mini.view(np.int16).copy())

    hdr = orc.header_pack(n_samples=SW * SH, cp=2, cb_height=18, cb_width=64, wl=SWL, bit_depth=8, lossy=0,
                          qs_1e4=10000, components=1, is_rgb=0, height=SH, endianess=0, bps=8, is_signed=0,
                          frames=0, k_1e3=0)
    full = pd.encode_frame_striped(sizes.size, encode_stripe, torch.from_numpy(hdr.view(np.int16).copy()), rank,
                                   world, torch.device("cpu"))
    if rank == 0:
        np.save(os.path.join(outdir, "striped.npy"), full.numpy().view(np.uint16))
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 5])
def test_codeblock_stripes_splice_to_single_gpu_stream(oracle, tmp_path, world):
    port = _free_port()
    mp.spawn(_stripe_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    ref = oracle.encode_frame(oracle.gen_frame(SW, SH, 9), SWL, False, 1.0, oracle.lut_for(False, SWL), 0, 0)
    assert np.array_equal(np.load(os.path.join(tmp_path, "striped.npy")), ref)


def test_stripe_ranges():
    sys.path.insert(0, os.path.join(ROOT, "cuda-image-and-video-codec_amd", "python"))
    import picsong_dist as pd
    assert pd.stripe_ranges(12, 5) == [(0, 3), (3, 3), (6, 2), (8, 2), (10, 2)]
    assert pd.stripe_ranges(3, 4) == [(0, 1), (1, 1), (2, 1), (3, 0)]
    assert sum(n for _, n in pd.stripe_ranges(65536, 8)) == 65536


# ---- intra-frame sharding WITH a row-band transform: the product's own kernels (CPU wave emulator) ----
BW, BH, BWL = 256, 256, 3          # 4 x 4 = 16 codeblocks; 2 ranks: 128 input rows each


def _banded_worker(rank, world, port, outdir, lossy):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.join(ROOT, "cuda-image-and-video-codec_amd", "python"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import emu_lib as E
    import oracle_lib as orc
    import picsong_dist as pd
    qs = 0.5 if lossy else 1.0
    lut = orc.lut_for(lossy, BWL)
    img = orc.pad_frame(orc.gen_frame(BW, BH, 11))
    AH, AW = img.shape
    P, extra = AW * AH, orc.dwt_extra(AW, AH, BWL)
    plan = pd.band_plan(AW, AH, world)
    me = plan[rank]
    # this rank holds ONLY its band and the 4-row halo of the input
    x = np.full((AH, AW), 0x5A, np.uint8)
    lo, hi = max(0, me["row0"] - 4), min(AH, me["row0"] + me["rows"] + 4)
    x[lo:hi] = img[lo:hi]
    x = E.aligned_copy(x)
    coef = E.aligned_zeros(P + extra, np.float32 if lossy else np.int32)
    coef[:] = 777                                    # what the rank never computes stays poison

    class Ops:
        def dwt_band(self, row0, rows):
            E.dwt_forward_band(x, coef, BWL, lossy, qs, row0, rows)

        def ll1(self):
            return torch.from_numpy(coef[P:P + (AW // 2) * (AH // 2)])      # a view: the gather lands in place

        def dwt_tail(self):
            E.dwt_forward_tail(coef, AW, AH, BWL, lossy, qs)

        def encode_stripe(self, b, n):
            staging, sizes, flag = E.bpc_encode_range(coef[:P].reshape(AH, AW), BWL, lut, b, n)
            assert flag == 0
            mini = E.pack(staging[b * 4096:(b + n) * 4096], sizes[b:b + n], None)
            return torch.from_numpy(mini.view(np.int16).copy())

    hdr = orc.header_pack(n_samples=BW * BH, cp=2, cb_height=18, cb_width=64, wl=BWL, bit_depth=8, lossy=int(lossy),
                          qs_1e4=int(qs * 10000), components=1, is_rgb=0, height=BH, endianess=0, bps=8, is_signed=0,
                          frames=0, k_1e3=0)
    full = pd.encode_frame_banded(AW, AH, Ops(), torch.from_numpy(hdr.view(np.int16).copy()), rank, world,
                                  torch.device("cpu"))
    if rank == 0:
        np.save(os.path.join(outdir, "banded.npy"), full.numpy().view(np.uint16))
    else:
        assert full is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("lossy", [False, True])
def test_banded_transform_and_stripes_splice_to_single_gpu_stream(oracle, tmp_path, lossy):
    """World size 2, the product's kernels through the wave emulator: each rank transforms its row band from
    an input that holds nothing else, the LL1 bands are all-gathered over gloo, each rank codes the two
    codeblock stripes its coefficients cover, rank 0 splices: the oracle's whole-frame codestream."""
    port = _free_port()
    mp.spawn(_banded_worker, args=(2, port, str(tmp_path), lossy), nprocs=2, join=True)
    qs = 0.5 if lossy else 1.0
    ref = oracle.encode_frame(oracle.gen_frame(BW, BH, 11), BWL, lossy, qs, oracle.lut_for(lossy, BWL), 0, 0)
    assert np.array_equal(np.load(os.path.join(tmp_path, "banded.npy")), ref)


def test_band_plan():
    sys.path.insert(0, os.path.join(ROOT, "cuda-image-and-video-codec_amd", "python"))
    import picsong_dist as pd
    p = pd.band_plan(16384, 16384, 8)                 # BASELINE config 5
    assert [q["rows"] for q in p] == [2048] * 8 and p[3]["row0"] == 6144
    assert p[1]["stripes"] == [(16 * 256, 16 * 256), ((128 + 16) * 256, 16 * 256)]
    covered = sorted(b for q in p for s in q["stripes"] for b in range(s[0], s[0] + s[1]))
    assert covered == list(range(65536))
    assert sum(q["ll1_count"] for q in p) == 8192 * 8192
    assert pd.band_plan(7680, 4352, 8) is None        # 4352 is not a multiple of 1024


# ---- bench.py's own launcher: `python bench.py --gpus N` starts the N ranks itself ---------
def _run_bench(*argv):
    import subprocess
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *argv], env=env, capture_output=True,
                          text=True, timeout=300)


def test_bench_launches_its_own_ranks_dry_gloo():
    """`python bench.py --gpus 2` without a torch.distributed environment spawns two ranks (a child
    torch.distributed.run), they rendezvous over gloo, run the per-step exchange on stand-in codestreams and rank 0's
    ONE JSON line comes back through the launcher with "n_gpus": 2."""
    import json
    r = _run_bench("--gpus", "2", "--backend", "gloo", "--dry", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["dry"] is True
    assert d["exchange"]["payloads_ok"] is True and d["scaling"] == "weak"


def test_bench_intra_frame_workload_dry_gloo():
    """BASELINE configs[4] has a driver-launchable form: `python bench.py --workload 16k_intra --gpus 2` starts two ranks
    that run picsong_dist.encode_frame_banded per step (LL1 all-gather, two gathers, splice).  Here with --dry over gloo:
    a stand-in codec whose mini-streams depend on a checksum of the all-gathered plane, so `splice_ok` holds only if
    every exchange moved what it should; "strong" scaling, one JSON line."""
    import json
    r = _run_bench("--workload", "16k_intra", "--gpus", "2", "--backend", "gloo", "--dry", "--steps", "3", "--warmup", "1")
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["dry"] is True and d["scaling"] == "strong"
    assert d["exchange"]["splice_ok"] is True and d["exchange"]["ranks_seen"] == 2


def test_bench_gpu_count_comes_from_sysfs_not_from_hip():
    """The launcher counts GPUs without opening the HIP runtime (a process that has touched the GPU must not fork + exec
    its ranks): KFD topology nodes with SIMDs, or None where there is no KFD (this container)."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    n = b.count_gpus_without_hip()
    assert n is None or (isinstance(n, int) and n >= 0)
    src = open(os.path.join(ROOT, "bench.py")).read()
    launch = src[src.index("def launch_ranks"):src.index("def rccl_witness")]
    assert "device_count" not in launch and "import torch" not in launch
    # the transform's byte counts of the line: S8(d)'s, and what the fused int16 design must move (never above it)
    P = 7680 * 4352
    assert abs(b.dwt_bytes(P, 5, 1) / 1e6 - 255.9) < 0.1 and abs(b.dwt_required_bytes(P, 5) / 1e6 - 122.2) < 0.1
    assert b.dwt_required_bytes(P, 5, False, False) == b.dwt_bytes(P, 5, 1)


def test_bench_refuses_more_gpus_than_the_node_has():
    """No quiet world = 1 run: asking for more GPUs than the node has is an error (here: none visible, or one)."""
    if torch.cuda.device_count() >= 64:
        pytest.skip("a node with 64 GPUs")
    r = _run_bench("--gpus", "64", "--steps", "1")
    assert r.returncode != 0
    assert "GPU(s)" in r.stderr and not r.stdout.strip()


def test_bench_refuses_a_world_size_that_contradicts_gpus():
    import subprocess
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--dry"], env=env,
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr
