#!/usr/bin/env python3
"""bench.py -- particle-scan evals/s of the SLAM hot path on MI355X (BASELINE.json metric).

One "step" = one LIDAR scan through the whole path, inputs already resident in HBM:
  set poses (motion-model samples are an input) -> score N particles against the likelihood field
  -> normalise / Neff / weighted pose -> resample if neff < N/2 -> ray-cast the scan into the
  log-odds map at the weighted pose -> rebuild the likelihood field where it changed.

Workload at 1 GPU = BASELINE.json configs[2] ("C3"): 16384 particles, 720 beams, 2048x2048 grid @ 2 cm, the
configuration the metric is quoted on.  At N > 1 GPUs the default is configs[3] ("C4"): a FIXED population of 65536
particles split over the ranks (8192 per GPU at 8; "scaling": "strong"), one grouped RCCL all-gather per scan (raw
weights + block partials), every rank keeping a replica of the map; the weak series (--config C3: 16384 particles per
GPU whatever N) rides along as secondary.weak.  --config C5 is the batched-map throughput mode (64 maps x 4096
particles x 1080 beams in one handle; at N GPUs every rank runs its own 64 maps, no collective); --config C2 the small
single-map case.

stdout carries ONE compact JSON line (< 4 KB: the contract's fields, roofline, cpu_baseline, a few figures beside them);
the full report -- per-kernel table, the secondary runs (C5, C2, C4 on one GPU, eight batched maps of the C3 shape, the C3
step on a map that is being explored, the closed loop, a soak, the replay of the shipped recording), notes -- goes to the
file the line names under "report" (--report, default bench_report.json beside this file).

  python bench.py --gpus 1 --steps 200 --warmup 20
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
         bench.py --gpus N --steps K --warmup W

value = total particles x steps / max-over-ranks wall time of the K steps.
roofline = the dominant kernel's algorithmic bytes per launch / its average launch duration: HIP events on the library's
stream around every launch of a REPLAY of the timed steps (the timed region itself carries no event brackets: they cost
stream time), minus what the two event markers add (calibrated on an empty kernel); roofline.step is the whole step's
algorithmic bytes over the wall time of a step.
cpu_baseline = the C oracle (a port of the reference's Java loops; the reference itself cannot run here) on a bounded
sample of the same workload, one host thread.  At N > 1 the run first VERIFIES itself: one untimed step on the sharded
filter and, on rank 0, on a stand-alone filter of the whole population; "sharded_equals_standalone" reports whether
every rank's particles, weights, statistics and map came out bit-identical.
"""
from __future__ import annotations

import argparse
import glob
import json
import math
import os
import sys
import time
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
def _microbench_ceilings():
    """Look-up ceilings of the scoring kernel, measured by tools/microbench/gather_coalesce.hip on MI355X and committed as
    profiles/<round>/microbench.json (tools/microbench/run_all.py; the newest round wins): independent 8-byte look-ups whose
    patch stays in L1, and the same when four neighbouring lanes share a line (particles in locality order)."""
    ind, quad, src = 818e9, 1590e9, "constants of round 2 (no profiles/*/microbench.json found)"
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "microbench.json"))):
        try:
            d = json.load(open(f))
            if d.get("gather_independent_lanes_per_s") and d.get("gather_neighbour_quads_lanes_per_s"):
                ind, quad, src = d["gather_independent_lanes_per_s"], d["gather_neighbour_quads_lanes_per_s"], os.path.relpath(f, ROOT)
        except Exception:
            pass
    return ind, quad, src


GATHER_CEILING_LANES_PER_S, GATHER_CEILING_MERGED_PER_S, GATHER_CEILING_SOURCE = _microbench_ceilings()


# ---------------------------------------------------------------------------------------------------------------------
# algorithmic bytes (SURVEY.md section 8(d) per-unit figures x the units one launch processes)
# ---------------------------------------------------------------------------------------------------------------------
def algorithmic_bytes(kind: str, *, n_particles=0, n_hit=0, n_beams=0, cells=0, visits=0, dirty_cells=0, n_maps=1,
                      paired=True, full_rebuild=False) -> float:
    """Per launch.  A paired launch (gms_fused_kernels.hip) does the work of both members; `paired` adds the partner's
    bytes to the class the launch is booked under (raycast: + normalise + the previous scan's apply;
    likelihood: + resample)."""
    score = 8.0 * n_particles * n_hit + 20.0 * n_particles + 17.0 * n_beams   # 8 B/beam-eval + pose+weight/particle + beam table
    reduce_ = 16.0 * n_particles                                              # 16 B per particle
    apply_ = 16.0 * visits                                                    # 16 B per visited cell (fp64 read + write)
    raycast = 16.0 * visits
    resample = 32.0 * n_particles                                             # 8 B weight + 12 B pose read + 12 B pose write
    lik = 16.0 * (cells if full_rebuild else dirty_cells)                     # 16 B per cell rebuilt
    order = 52.0 * n_particles                                                # 12 B pose read, 20 B pose+trig and 20 B ordered copy + index written
    # (the previous scan's apply pass rides in the ray-cast launch of a paired step: beside the partials until round 2's second count grid)
    per_map = {"score": score, "reduce": reduce_, "apply": apply_,
               "raycast": raycast + ((reduce_ + apply_) if paired else 0.0), "likelihood": lik + (resample if paired else 0.0),
               "resample": resample, "order": order}.get(kind, 0.0)
    return per_map * n_maps


def pmc_traffic(config: str, kernel_class: str):
    """HBM-side bytes per launch from the committed PMC passes (profiles/*/pmc_traffic*.json: separate
    `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` runs of this command, FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes for gfx950); None when not collected for this configuration.  The newest round wins."""
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "pmc_traffic*.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        d = d.get(config, d if config == "C3" and "score" in d else {})      # r01's file is C3 only, un-keyed
        if kernel_class in d:
            best = d[kernel_class].get("hbm_bytes_per_launch")
    return best


# ---------------------------------------------------------------------------------------------------------------------
class Workload:
    """One BASELINE configuration on this rank: map pre-built from the first half of a synthetic trace, pose sets
    resident in HBM, the filter (stand-alone, or this rank's shard)."""

    def __init__(self, name, args, torch, dist, rank, world, local_rank, sharded, keep_log=False, loop=False, comm=None,
                 explore=False, n_maps=None):
        from gridmap_slam_robot_amd import GridMap, ParticleFilter, synth
        self.name, self.args, self.torch, self.dist = name, args, torch, dist
        self.rank, self.world = rank, world
        self.loop = loop        # closed loop: the particles are what the step before left, moved by the motion model on the device
        self.explore = explore  # the map starts EMPTY and the timed scans are the drive's first ones: the frontier moves every step
        cfg = dict(synth.CONFIGS[name])
        self.M = n_maps or cfg["n_maps"]
        self.batched = self.M > 1
        # config 5 at N > 1: its 64 maps are split over the ranks, 64 / N independent maps per GPU, no collective on the
        # data path: a FIXED job, strong scaling -- the one BASELINE configuration whose work divides without a replicated part
        self.map_offset = 0
        self.maps_total = self.M
        if self.batched and world > 1 and n_maps is None:
            if self.M % world:
                raise SystemExit(f"bench.py: {self.M} maps do not split over {world} ranks")
            self.M //= world
            self.map_offset = rank * self.M
            self.batched = True              # (one map per rank still runs as a batched handle's arithmetic would: the same kernels)
        self.sharded = sharded and not self.batched
        # C4 is BASELINE.json configs[3]: a FIXED population of 65 536 particles split over the ranks (strong scaling; --particles
        # overrides the population); every other configuration keeps the per-GPU work fixed as N grows (weak scaling; --particles
        # overrides the particles per GPU and per map)
        self.strong = name == "C4" or (self.M != self.maps_total)
        if name == "C4":
            total = args.particles or cfg["particles"]
            parts = world if self.sharded else 1
            if total % parts or (parts > 1 and (total // parts) % 256):
                raise SystemExit(f"bench.py: {total} particles do not split into {parts} shards of a multiple of 256")
            self.n_local = total // parts
        else:
            self.n_local = args.particles or cfg["particles"]
        self.n_global = self.n_local * (world if self.sharded else 1)
        self.B, self.ext, self.res = cfg["beams"], cfg["extent"], cfg["resolution"]
        self.T = T = 64 if not self.batched else 16
        self.dev = dev = torch.device("cuda", local_rank)
        self.n_sets = 8 if not self.batched else 4
        ext, res, B, M = self.ext, self.res, self.B, self.M
        self.synth = synth
        if not self.batched:
            self.tr = tr = synth.make_trace(ext, res, B, T=T, seed=1234)
            self.m = m = GridMap(ext, ext, res, (-ext / 2, -ext / 2), device=local_rank, max_beams=max(2048, B))
            self.stream = torch.cuda.current_stream()
            m.set_stream(self.stream.cuda_stream)
            for t in range(T // 2):                  # pre-built map: likelihoods are informative
                m.update(tr.scans[t], tr.poses[t])
            m.synchronize()
            self.log0 = m.download_log().reshape(-1).copy() if keep_log else None
            self.scans_dev = torch.from_numpy(tr.scans.view(np.uint8).reshape(T, -1).copy()).to(dev)
            self.pose_sets, self.pose_sets_host = [], []
            for s in range(self.n_sets):
                allp = synth.make_particles(tr.poses[T // 2 + s], self.n_global, seed=99 + s)      # sigma 0.10 m / 5 deg
                mine = allp[rank * self.n_local:(rank + 1) * self.n_local] if self.sharded else allp
                self.pose_sets.append(torch.from_numpy(np.ascontiguousarray(mine)).to(dev))
                self.pose_sets_host.append(np.ascontiguousarray(mine))
                if s == 0:
                    self.global_set0 = allp
            self.n_hit = int(tr.scans[T // 2]["hit"].sum())
            self.scan0 = tr.scans[T // 2]
            self.pose0 = tr.poses[T // 2]
        else:
            nt = min(M, 8)
            self.traces = trs = [synth.make_trace(ext, res, B, T=T, seed=100 + i) for i in range(nt)]
            self.m = m = GridMap(ext, ext, res, (-ext / 2, -ext / 2), n_maps=M, device=local_rank, max_beams=max(2048, B))
            self.stream = torch.cuda.current_stream()
            m.set_stream(self.stream.cuda_stream)
            mo = self.map_offset                 # this rank's maps are maps [mo, mo + M) of the job
            for t in range(T // 2):
                m.update(np.stack([trs[(mo + i) % nt].scans[t] for i in range(M)]), np.stack([trs[(mo + i) % nt].poses[t] for i in range(M)]))
            m.synchronize()
            self.log0 = None
            self.scans_dev = [torch.from_numpy(np.stack([trs[(mo + i) % nt].scans[t] for i in range(M)]).view(np.uint8).copy()).to(dev)
                              for t in range(T)]
            self.pose_sets = []
            for s in range(self.n_sets):
                t = T // 2 + s
                P = np.stack([synth.make_particles(trs[(mo + i) % nt].poses[t], self.n_local, seed=7 + mo + i + 64 * s) for i in range(M)])
                self.pose_sets.append(torch.from_numpy(P).to(dev))
            self.n_hit = int(trs[0].scans[T // 2]["hit"].sum())
            self.scan0 = trs[0].scans[T // 2]
            self.pose0 = trs[0].poses[T // 2]
        self.r01 = np.ascontiguousarray(np.random.default_rng(7).random((4096, M)))
        self.pose_ptrs = [p.data_ptr() for p in self.pose_sets]         # (raw device addresses, taken once: the hot loop passes integers)
        self.beam_ptrs = [x.data_ptr() for x in self.scans_dev]

        self.spf = self.comm = self.ops = None
        self.route = None
        self.two_collectives = False
        if self.sharded:
            from gridmap_slam_robot_amd.distributed import HipShardOps, RcclComm, ShardedParticleFilter
            self.ops = HipShardOps(m, self.n_local, rank * self.n_local, self.n_global)
            self.stream = self.ops.stream
            self.spf = ShardedParticleFilter(self.n_global, self.ops)
            self.pf = self.ops.pf
            if comm is not None:
                self.comm = comm                  # (one communicator per device: a second workload of the run shares the first's)
            elif args.exchange in ("auto", "in-library"):
                ok = 1
                try:
                    self.comm = RcclComm(local_rank)
                except Exception as e:            # keep the run alive: the torch.distributed exchange does the same job
                    print(f"bench.py: in-library RCCL communicator unavailable ({e}); using torch.distributed", file=sys.stderr)
                    ok = 0
                if not self.agree(ok) and self.comm is not None:
                    self.comm.close()
                    self.comm = None
        else:
            self.pf = ParticleFilter(m, self.n_local)
        if self.loop:
            # the filter starts one odometry step before the first timed scan, tightly around the true pose; from then on its
            # particles are its own: resampled by the step, moved by gms_pf_sample_motion (Odometry.apply, Odometry.java:77-96)
            t0 = self.T // 2 - 1
            self.pf.set_poses(synth.make_particles(self.tr.poses[t0], self.n_local, seed=5, sigma_xy=0.02, sigma_theta_deg=1.0))
            d = self.tr.poses[t0 + 1].astype(np.float64) - self.tr.poses[t0].astype(np.float64)
            self.odo = (float(math.hypot(d[0], d[1])), float(d[2]))          # the drive is a circle: every frame's odometry is the same

    # -- helpers ------------------------------------------------------------------------------------------------------
    def agree(self, ok: int) -> int:
        """every rank must take the same path"""
        if self.world > 1 and self.dist.is_initialized():
            t_ok = self.torch.tensor([ok], dtype=self.torch.int32, device=self.dev)
            self.dist.all_reduce(t_ok, op=self.dist.ReduceOp.MIN)
            return int(t_ok.item())
        return ok

    def barrier(self):
        # The host polls an event on the library's stream first, so that the synchronize below finds the queue drained instead of
        # sleeping on it: a blocking wait wakes up tens of microseconds late, which at the driver's 20 steps (1 ms in all) is a few
        # per cent of the timed region.  The synchronize + barrier + synchronize bracket itself is unchanged.
        ev = self.torch.cuda.Event()
        ev.record(self.stream)
        while not ev.query():
            pass
        self.torch.cuda.synchronize()
        if self.world > 1 and self.dist.is_initialized():
            self.dist.barrier()
            self.torch.cuda.synchronize()

    def beams_ptr(self, t):
        return self.beam_ptrs[t]

    # -- one scan step ------------------------------------------------------------------------------------------------
    def step(self, i: int):
        a, pf, m, B = self.args, self.pf, self.m, self.B
        if self.loop:
            t = (self.T // 2 + i) % self.T                      # the drive goes on round the circle, scan after scan
            # SLAM.java:87-131 with :90's motion-model sample inside the scoring launch, GridMapApp.java:185-186
            pf.slam_update_u_dev(self.odo[0], self.odo[1], 7, i, self.beams_ptr(t), B, float(self.r01[i % 4096][0]), 0.5, True)
            return
        s = i % self.n_sets
        t = self.T // 2 + s
        bp = self.beams_ptr(t)
        r01 = self.r01[i % 4096]
        if self.batched:
            pf.slam_update_dev(self.pose_sets[s].data_ptr(), bp, B, r01, 0.5, True)      # one C-ABI call per batched scan
            return
        r01v = r01                                           # (the [n_maps] row itself: no conversion on the hot path)
        r01 = float(r01[0])
        if a.host_inputs and self.spf is None:
            pf.slam_update(self.pose_sets_host[s], self.tr.scans[t], r01, 0.5, True)
            return
        if self.spf is None and not a.full_rebuild:
            pf.slam_update_dev(self.pose_ptrs[s], bp, B, r01v, 0.5, True)                 # one C-ABI call per scan
            return
        if self.comm is not None:
            pf.slam_update_sharded_dev(self.comm, self.pose_sets[s].data_ptr(), bp, B, r01, 0.5, True)
            return
        if self.spf is not None and not self.two_collectives:
            # the same single-exchange step with the two all-gathers issued through torch.distributed
            self.spf.scan_step((self.pose_sets[s].data_ptr(), bp, B, True), r01, 0.5)
            return
        if self.spf is not None:
            # last resort: all-reduce of the partials, all-gather of the normalised particles (distributed.py)
            pf.set_poses_dev(self.pose_sets[s].data_ptr())
            pf.score_dev(bp, B)
            self.spf.normalize_begin()
            m.update_at_dev(bp, B, pf)
            self.spf.normalize_end()
            self.spf.resample(r01, 0.5)
            return
        # --full-rebuild: the separate entry points, likelihood field rebuilt everywhere as the reference does
        pf.set_poses_dev(self.pose_sets[s].data_ptr())
        pf.score_dev(bp, B)
        pf.normalize(fetch=False)
        pf.resample_if(r01, 0.5)
        m.integrate_at_dev(bp, B, pf)
        m.compute_likelihood_map()

    # -- exchange route (sharded runs): one untimed step; if a route cannot run here every rank falls back together ---
    def pick_route(self):
        if self.spf is None:
            return
        routes = ["in-library", "torch-single", "torch-two"]
        if self.args.exchange != "auto":
            routes = routes[routes.index(self.args.exchange):]
        for route in routes:
            if route == "in-library" and self.comm is None:
                continue
            if route == "torch-single":
                self.comm = None
            if route == "torch-two":
                self.comm, self.two_collectives = None, True
            ok = 1
            try:
                self.step(0)
                self.torch.cuda.synchronize()
            except Exception as e:
                print(f"bench.py: sharded step via {route} failed ({e})", file=sys.stderr)
                ok = 0
            if self.agree(ok):
                self.route = route
                return
        raise RuntimeError("no exchange route works on this node")

    def exchange_text(self):
        if self.spf is None:
            return None
        return {"in-library": "in-library RCCL: one grouped all-gather per scan (raw weights + block partials)",
                "torch-single": "torch.distributed (RCCL), one exchange: two all-gathers issued together",
                "torch-two": "torch.distributed (RCCL), all-reduce of the partials + all-gather of the normalised particles"}[self.route]

    # -- N > 1: is the sharded filter the stand-alone filter? ---------------------------------------------------------
    def verify_against_standalone(self):
        """One untimed step on the sharded filter (every rank) and on a stand-alone filter of the whole population
        (rank 0, its own map replica copied from the pre-built one).  Every rank reports its particles, weights,
        statistics and map checksums; rank 0 compares them with the stand-alone filter's, bit for bit."""
        from gridmap_slam_robot_amd import GridMap, ParticleFilter
        torch, dist = self.torch, self.dist
        ext, res, B = self.ext, self.res, self.B
        t0 = self.T // 2
        r01 = float(self.r01[0][0])
        # Every rank reaches the one collective of this check (all_gather_object) whatever happens locally: a rank that
        # failed reports the error instead of its results, so a local failure cannot leave the ranks' collectives mismatched.
        ref, ref_err = None, None
        if self.rank == 0:
            try:
                m2 = GridMap(ext, ext, res, (-ext / 2, -ext / 2), device=self.dev.index, max_beams=max(2048, B))
                m2.set_stream(self.stream.cuda_stream)
                m2.copy_from(self.m)
                pf2 = ParticleFilter(m2, self.n_global)
                P = torch.from_numpy(np.ascontiguousarray(self.global_set0)).to(self.dev)
                with torch.cuda.stream(self.stream):
                    pf2.slam_update_dev(P.data_ptr(), self.beams_ptr(t0), B, r01, 0.5, True)
                torch.cuda.synchronize()
                ref = dict(stats=pf2.stats(), poses=pf2.get_poses(), weights=pf2.get_weights(),
                           log_crc=zlib.crc32(m2.download_log().tobytes()), lik_crc=zlib.crc32(m2.download_likelihood().tobytes()))
                pf2.close(); m2.close()
            except Exception as e:
                ref_err = repr(e)
        self.step(0)                                    # set 0, scan t0, r01[0]: the same inputs (collective inside: every rank)
        torch.cuda.synchronize()
        rccl_ranks = None
        try:
            mine = dict(rank=self.rank, stats=self.pf.stats(), poses=self.pf.get_poses(), weights=self.pf.get_weights(),
                        log_crc=zlib.crc32(self.m.download_log().tobytes()), lik_crc=zlib.crc32(self.m.download_likelihood().tobytes()))
            if self.comm is not None:
                import ctypes as C
                from gridmap_slam_robot_amd import _lib
                r, w = C.c_int32(), C.c_int32()
                _lib.check(_lib.load().gms_comm_rank(self.comm._h, C.byref(r), C.byref(w)))
                rccl_ranks = int(w.value)
                mine["comm_rank"] = int(r.value)
        except Exception as e:
            mine = dict(rank=self.rank, error=repr(e))
        if self.world > 1:
            box = [None] * self.world
            dist.all_gather_object(box, mine)
        else:
            box = [mine]
        if self.rank != 0:
            return None
        if ref is None:
            return {"sharded_equals_standalone": None, "error": f"stand-alone filter on rank 0: {ref_err}", "rccl_ranks": rccl_ranks,
                    "route_verified": self.route, "population": self.n_global}
        n = self.n_local
        detail = {}
        ok = True
        for e in box:
            r = e["rank"]
            if "error" in e:
                ok = False
                detail[f"rank{r}"] = [e["error"]]
                continue
            checks = dict(stats=e["stats"] == ref["stats"],
                          poses=bool(np.array_equal(e["poses"], ref["poses"][r * n:(r + 1) * n])),
                          weights=bool(np.array_equal(e["weights"], ref["weights"][r * n:(r + 1) * n])),
                          log=e["log_crc"] == ref["log_crc"], likelihood=e["lik_crc"] == ref["lik_crc"])
            if "comm_rank" in e:
                checks["comm_rank"] = e["comm_rank"] == r
            if not all(checks.values()):
                ok = False
                detail[f"rank{r}"] = [k for k, v in checks.items() if not v]
        return {"sharded_equals_standalone": ok, "mismatches": detail or None, "rccl_ranks": rccl_ranks,
                "route_verified": self.route, "population": self.n_global}

    # -- what one scan touches (algorithmic bytes of the map kernels) -------------------------------------------------
    def scan_footprint(self):
        """visited cells of one scan and the cells of the likelihood tiles a dirty rebuild covers (map 0)."""
        from gridmap_slam_robot_amd import GridMap
        probe = self.m
        if self.batched:        # trace_scan works on map 0 of a handle: a one-map handle of the same geometry
            probe = GridMap(self.ext, self.ext, self.res, (-self.ext / 2, -self.ext / 2), device=self.dev.index, max_beams=max(2048, self.B))
        cells, cls, counts = probe.trace_scan(self.scan0, self.pose0)
        visits = int(counts.sum())
        k = (probe.params.ktaps - 1) // 2
        sel = np.zeros(cls.shape, dtype=bool)
        for b in range(len(counts)):
            sel[b, :counts[b]] = cls[b, :counts[b]] != 1
        if sel.any():
            xs, ys = cells[..., 0][sel], cells[..., 1][sel]
            x0, x1 = max(int(xs.min()) - k, 0), min(int(xs.max()) + k, probe.W - 1)
            y0, y1 = max(int(ys.min()) - k, 0), min(int(ys.max()) + k, probe.H - 1)
            dirty = (x1 // 64 - x0 // 64 + 1) * (y1 // 32 - y0 // 32 + 1) * 64 * 32
        else:
            dirty = 0
        if probe is not self.m:
            probe.close()
        return visits, int(dirty)


# ---------------------------------------------------------------------------------------------------------------------
def preroll_steps() -> int:
    try:
        return max(0, int(os.environ.get("GMS_BENCH_PREROLL", "300")))
    except ValueError:
        return 300


def timed_regions() -> int:
    try:
        return max(1, int(os.environ.get("GMS_BENCH_REGIONS", "5")))
    except ValueError:
        return 5


def measure(wl: Workload, steps: int, warmup: int):
    """warm-up (untimed), the timed region (exactly `steps` steps, NO event brackets: nothing but the steps themselves
    between the two barriers), then a bracketed replay of the same steps: every launch of every kernel class between
    HIP events on the library's stream, for the roofline and the per-kernel table."""
    m, torch, dist = wl.m, wl.torch, wl.dist
    bracket_ms, noop_ms = m.profile_calibrate2(200)
    m.profile(False)
    # An untimed pre-roll in front of the warm-up: a timed region of the driver's 20 steps is one millisecond long and starts 0.3 ms
    # after the process's first scan step -- clocks still ramping, first-use allocations of the runtime still ahead
    # (GMS_BENCH_PREROLL=0 turns it off; the count is in the report).
    # ... and in front of THAT, the protocol of rounds 1-4 as it was: `warmup` steps, then ONE region of `steps` steps, timed cold -- the
    # process's first scan steps.  Reported as first_region_ms_per_step so that rounds stay comparable; never `value`.
    cold = None
    if not wl.loop:
        for i in range(warmup):
            wl.step(i)
        wl.barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            wl.step(warmup + i)
        wl.barrier()
        cold = time.perf_counter() - t0
        if wl.world > 1 and dist.is_initialized():
            tt = torch.tensor([cold], dtype=torch.float64, device=wl.dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            cold = float(tt.item())
        for i in range(preroll_steps()):
            wl.step(i % max(warmup + steps, 1))
        wl.barrier()
    import gc
    gc.collect()
    gc.disable()                              # (a collection inside a one-millisecond region is half of it: one process in ten read 70 us per step)
    try:
        return _measure_timed(wl, steps, warmup, bracket_ms, noop_ms, cold)
    finally:
        gc.enable()


def _measure_timed(wl: Workload, steps: int, warmup: int, bracket_ms: float, noop_ms: float, cold):
    m, torch, dist = wl.m, wl.torch, wl.dist
    for i in range(warmup):
        wl.step(i)
    # The timed region -- exactly `steps` steps between two barriers -- is taken timed_regions() times over, back to back, and the
    # MEDIAN region is the one reported (all of them are in the report): at the driver's 20 steps a region is one millisecond long,
    # and about one process in ten had a 0.3-0.5 ms stall of unknown origin land in it (tools/bench20_spread.sh: 47 us per step read
    # as 61-74).  Every region is K steps of the same workload; at N > 1 a region's time is the maximum over the ranks.
    regions = []
    for r in range(timed_regions()):
        base = warmup + (r * steps if wl.loop else 0)
        wl.barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            wl.step(base + i)
        issue_r = time.perf_counter() - t0    # host time to enqueue K steps (the GPU must not be waiting on it)
        wl.barrier()
        el = time.perf_counter() - t0
        pr = [el]
        if wl.world > 1 and dist.is_initialized():
            tt = torch.tensor([el], dtype=torch.float64, device=wl.dev)
            allt = [torch.zeros_like(tt) for _ in range(wl.world)]
            dist.all_gather(allt, tt)
            pr = [float(x.item()) for x in allt]
        regions.append((max(pr), issue_r, pr))
    import gc
    gc.enable()
    elapsed, issue, per_rank = sorted(regions, key=lambda x: x[0])[len(regions) // 2]
    if wl.loop:
        warmup += (len(regions) - 1) * steps   # (a closed loop's replay below goes on from where the last region ended)

    # bracketed replay: the same steps again (the same pose sets, scans and draws), every launch timed
    m.profile(True)
    m.profile_reset()
    # (a short timed region is replayed several times over, to at least 100 steps: the first bracketed launches behind a barrier run
    # cold and, at the driver's 20 steps, would be a fifth of the sample)
    nb = max(1, min(max(steps, 100 if not wl.batched else 20), 200))
    first = warmup + (steps if wl.loop else 0)           # a closed loop cannot go back: its replay is the NEXT nb steps of the drive
    for i in range(nb):
        wl.step(first + i)
    wl.barrier()
    prof = m.profile_get()
    m.profile(False)
    # the steady-state step, un-bracketed: what the bracketed readings are calibrated against (the timed region of a short run
    # carries ~50 us of first-launch and completion latency that belongs to no kernel: 2.5 us per step at 20 steps)
    ns = max(nb, 200 if not wl.batched else 40)
    wl.barrier()
    t1 = time.perf_counter()
    for i in range(ns):
        wl.step(first + nb + i)
    wl.barrier()
    steady = (time.perf_counter() - t1) / ns
    # census of the likelihood tiles over nb more steps (counters in the rebuild's workgroups, off everywhere else): what the
    # dirty-tile rebuilds of this workload rewrite and what they leave alone
    tiles = None
    try:
        m.tile_stats(True, fetch=False)
        for i in range(nb):
            wl.step(first + nb + ns + i)
        wl.barrier()
        tiles = {k: v / nb for k, v in m.tile_stats(False).items()}
    except Exception as e:
        print(f"bench.py: tile census unavailable: {e!r}", file=sys.stderr)
    compute = [k for k in prof if prof[k][1] > 0 and k != "exchange"]
    dominant = max(compute, key=lambda k: prof[k][0], default="score")
    return dict(elapsed=elapsed, issue=issue, per_rank=per_rank, dominant=dominant, prof=prof, nb=nb, steady=steady, steady_steps=ns, tiles=tiles,
                preroll=0 if wl.loop else preroll_steps(), regions=[x[0] for x in regions], cold_region=cold,
                warmup_effective=warmup if wl.loop else warmup + steps + preroll_steps() + warmup,
                bracket_us=bracket_ms * 1e3, noop_us=noop_ms * 1e3)


def report(wl: Workload, meas: dict, steps: int, warmup: int):
    """the JSON fields of one configuration (rank 0)."""
    a = wl.args
    visits, dirty = wl.scan_footprint()
    m = wl.m
    n_total = wl.n_global * wl.M * (wl.world if (wl.batched and wl.world > 1) else 1)
    elapsed = meas["elapsed"]
    value = n_total * steps / elapsed
    paired = not a.full_rebuild and not a.host_inputs or wl.batched
    # the likelihood rebuild is charged for the cells it REWRITES (16 B each: log-odds read, factor written): the census' written and
    # blurred tiles; tiles the unchanged-tile rule leaves alone, and uniform tiles that already hold their constants, move nothing
    # that the algorithm needs (their staging reads are overhead, not algorithmic bytes)
    tiles = meas.get("tiles")
    dirty_box_cells = dirty
    if tiles is not None and not a.full_rebuild:
        dirty = int(round((tiles["constants_written"] + tiles["blurred"]) * 64 * 32 / wl.M))
    kw = dict(n_particles=wl.n_local, n_hit=wl.n_hit, n_beams=wl.B, cells=m.W * m.H, visits=visits, dirty_cells=dirty, n_maps=wl.M,
              paired=bool(paired), full_rebuild=a.full_rebuild)
    nb, prof = meas["nb"], meas["prof"]
    # The two event markers of a bracket are stream commands of their own, so a bracketed launch reads longer than the launch
    # costs in the un-bracketed step.  How much longer is taken from the run itself: the bracketed readings of one step add up to
    # more than an un-bracketed step takes (the steady-state step: the timed region's, or that of a longer un-bracketed run right after
    # the replay when the timed region is short), and the excess, spread over the step's launches, is what a bracket adds (marker_us).  avg_launch_us = reading - marker_us: per-launch durations that TILE the timed step -- launch overhead and
    # the gap to the next kernel included, which is also what `rocprofv3 --kernel-trace --stats` attributes to a kernel
    # (profiles/: k_score_c 20.0 us there).  The start-up calibration on empty kernels (gms_profile_calibrate2: a bracket around
    # an empty kernel minus an empty kernel back to back, ~5.1 us) overstates it: behind an empty kernel nothing overlaps the
    # markers' processing; it is kept as the upper bound of the correction and reported (event_markers_empty_kernel_us).
    marker_empty_us = max(0.0, meas["bracket_us"] - meas["noop_us"])
    launches_per_step = sum(n for _, n in prof.values()) / nb
    raw_us_per_step = sum(ms for ms, _ in prof.values()) / nb * 1e3
    step_us = min(elapsed / steps, meas.get("steady", elapsed / steps)) * 1e6      # the steady-state step (see measure)
    marker_us = min(max(0.0, (raw_us_per_step - step_us) / max(launches_per_step, 1.0)), marker_empty_us)
    # Long launches (C5's scoring launch: 210 us between markers, 220-230 in the un-bracketed step, 240 under rocprofv3) show the
    # opposite: the bracketed readings of a step add up to LESS than the un-bracketed step, the pauses the markers insert letting the
    # kernels run faster than they do back to back.  Then every reading is scaled up by the same factor, so that the durations
    # still tile the timed step (tile_scale; 1.0 whenever the additive correction applies).
    tile_scale = step_us / raw_us_per_step if (0.0 < raw_us_per_step < step_us and not a.host_inputs) else 1.0
    kernels = {}
    for k, (ms, n) in prof.items():
        if not n:
            continue
        lps = n / nb
        ab = algorithmic_bytes(k, **kw) / max(1.0, lps) if k != "exchange" else None
        raw_us = ms / n * 1e3
        us = max(raw_us - marker_us, 0.05) * tile_scale
        tr = pmc_traffic(wl.name, k) if not (a.full_rebuild or a.particles or a.host_inputs) else None
        e = {"launches_per_step": round(lps, 2), "avg_launch_us": round(us, 2), "avg_bracketed_us": round(raw_us, 2),
             "us_per_step": round(us * lps, 2)}
        if ab:
            e.update({"algorithmic_bytes_per_launch": int(ab), "algorithmic_gb_per_s": round(ab / (us * 1e-6) / 1e9, 1),
                      "hbm_frac": round(ab / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4)})
            if tr:
                e.update({"traffic_bytes_per_launch": int(tr), "traffic_over_algorithmic": round(tr / ab, 2)})
        kernels[k] = e
    dom = meas["dominant"]
    lps_dom = max(1.0, prof[dom][1] / nb) if prof[dom][1] else 1.0
    alg = algorithmic_bytes(dom, **kw) / lps_dom
    dom_raw_s = (prof[dom][0] / max(prof[dom][1], 1)) * 1e-3
    dom_avg_s = max(dom_raw_s - marker_us * 1e-6, 5e-8) * tile_scale
    achieved = alg / dom_avg_s / 1e9 if dom_avg_s > 0 else 0.0
    # the whole step against the same peak: SURVEY 8(d)'s per-unit figures over everything one scan does (8 B per beam
    # evaluation + pose, weight and beam table; 16 B per visited cell of the map update; 16 B per cell of the likelihood
    # rebuild; 16 B per particle for the normaliser, 32 B per particle for the resampling) over the wall time of a step
    step_bytes = (algorithmic_bytes("score", **kw) + algorithmic_bytes("apply", **kw) + algorithmic_bytes("reduce", **kw)
                  + algorithmic_bytes("resample", **kw) + algorithmic_bytes("likelihood", **dict(kw, paired=False)))
    step_s = elapsed / steps
    roof = {
        "kernel": dom, "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "traffic": pmc_traffic(wl.name, dom) if not (a.full_rebuild or a.particles or a.host_inputs) else None,
        "algorithmic_bytes_per_launch": alg, "avg_launch_us": dom_avg_s * 1e6, "avg_bracketed_us": dom_raw_s * 1e6,
        "launches_timed": prof[dom][1],
        "measured": f"HIP events on the library's stream around every launch of a replay of the timed steps ({nb} steps), minus what "
                    "the two event markers of a bracket add (event_markers_us: the excess of one step's bracketed readings over the "
                    "un-bracketed step of the timed region, per launch; when the readings add up to less than that step they are "
                    "scaled by tile_scale instead), so that the per-launch durations tile the timed step as a kernel trace's do",
        "event_markers_us": marker_us, "tile_scale": tile_scale, "event_markers_empty_kernel_us": marker_empty_us,
        "steady_state_step_us": step_us, "steady_state_steps": meas.get("steady_steps"),
        "event_bracket_empty_kernel_us": meas["bracket_us"], "empty_kernel_back_to_back_us": meas["noop_us"],
        "step": {"algorithmic_bytes_per_step": step_bytes, "achieved": step_bytes / step_s / 1e9, "unit": "GB/s",
                 "frac": step_bytes / step_s / 1e9 / HBM_PEAK_GBS, "launches_per_step": round(sum(v[1] for v in prof.values()) / nb, 2),
                 "kernel_us_per_step": round(sum(e["us_per_step"] for e in kernels.values()), 2)},
    }
    if roof["traffic"] and roof["traffic"] < 0.5 * alg:
        # the HBM peak is the yardstick the contract asks for, not what binds: most of the algorithmic bytes never leave the
        # caches, so `frac` can pass 1 (C5 with the particles in locality order)
        roof["note"] = ("%.0f %% of the algorithmic bytes reach the fabric (PMC): the working set is served by L2 / Infinity Cache, "
                        "frac is algorithmic bytes over the HBM peak and may exceed 1" % (100.0 * roof["traffic"] / alg))
    symbols = dict(KERNEL_SYMBOLS)
    if wl.batched:
        symbols.update({"raycast": "k_raycast_tile", "reduce": "k_partials + k_normalize_pack"})
    elif a.full_rebuild:
        symbols.update({"raycast": "k_raycast", "likelihood": "k_likelihood"})
    if wl.spf is not None:
        symbols.update({"raycast": "k_raycast_norm_chunks", "reduce": "k_partials_pack_apply"})
    roof["kernel_symbol"] = symbols.get(dom, dom)
    if roof["traffic"]:
        roof["traffic_over_algorithmic"] = roof["traffic"] / alg
    if dom == "score":
        # ONE pair of yardsticks for every configuration: `frac` is algorithmic bytes over the HBM peak (the contract's figure;
        # it passes 1 when the factor tables never leave L2 / Infinity Cache, as at C5), and `lookup_ceiling_frac` is look-ups
        # per second over the measured ceiling of the pipe that does bind the kernel (tools/microbench/gather_coalesce.hip: the L1
        # address pipe takes ~48 clocks per 64-lane 8-byte gather of independent lines, ~25 when neighbouring lanes share
        # lines -- the ceiling that applies when k_order has put the particles in locality order).
        roof["bound_measured"] = "l1-gather (texture-address pipe: ~48 clocks per 64-lane look-up, ~25 when neighbouring lanes share lines)"
        lookups = wl.n_local * wl.n_hit * wl.M
        ordered = bool(prof.get("order", (0, 0))[1])
        ceil = GATHER_CEILING_MERGED_PER_S if ordered else GATHER_CEILING_LANES_PER_S
        roof["lookups_per_s"] = lookups / dom_avg_s
        roof["lookup_ceiling_per_s"] = ceil
        roof["lookup_ceiling_frac"] = lookups / dom_avg_s / ceil
        roof["particles_in_locality_order"] = ordered
        roof["gather_ceilings"] = {"independent_lanes_per_s": GATHER_CEILING_LANES_PER_S, "neighbour_quads_lanes_per_s": GATHER_CEILING_MERGED_PER_S,
                                   "source": GATHER_CEILING_SOURCE}
    st = wl.pf.stats()
    st0 = st[0] if isinstance(st, list) else st
    # Neff recomputed on the host from the LOG-weights of the same scored population (no underflow there): if the raw
    # product's underflow cost the filter anything, the two would differ.  They do not: the particles whose product
    # underflows carry < 1e-150 of the weight.  (A tempered likelihood would keep more particles alive; the reference
    # has none, and none is added.)
    neff_log = None
    if wl.spf is None:                      # (a shard holds only its own log-weights)
        lw = np.asarray(wl.pf.get_log_weights(), dtype=np.float64).reshape(wl.M, -1)[0]
        sm = np.exp(lw - lw.max())
        neff_log = float(sm.sum() ** 2 / (sm * sm).sum())
    out = {
        "ms_per_step": elapsed / steps * 1e3,
        "timed_region_s": elapsed,
        "preroll_steps": meas.get("preroll", 0),            # untimed, in front of the warm-up (measure)
        "timed_regions": len(meas.get("regions", [1])),     # regions of `steps` steps timed back to back; ms_per_step is the median one's
        "region_ms_per_step": [round(x / steps * 1e3, 6) for x in meas.get("regions", [])],
        # the protocol of rounds 1-4 (W warm-up steps, then one region of K steps, the process's first): for comparison across rounds
        "first_region_ms_per_step": (round(meas["cold_region"] / steps * 1e3, 6) if meas.get("cold_region") else None),
        "warmup_effective": meas.get("warmup_effective", warmup),   # untimed steps in front of the reported regions: warm-up + cold region + pre-roll + warm-up
        "host_issue_ms_per_step": meas["issue"] / steps * 1e3,
        "config": {
            "workload": (f"{wl.name}: {wl.M} maps x {m.W}x{m.H} @ {wl.res} m x {wl.n_local} particles x {wl.B} beams ({wl.n_hit} hits), batched handle, "
                         if wl.batched else
                         f"{wl.name}: {wl.n_local} particles/GPU x {wl.B} beams ({wl.n_hit} hits), {m.W}x{m.H} grid @ {wl.res} m, ")
                        + "full scan step (score+normalise+resample+ray-cast+likelihood rebuild"
                        + (": every cell, likelihoodData written, as the reference does)" if a.full_rebuild else
                           "; the rebuild is dirty-tile / factor-table-only: bit-identical results, likelihoodData on demand -- the reference's own "
                           "every-cell rebuild is cpu_baseline.gpu_like_for_like_ms_per_step)"),
            "particles_total": n_total, "particles_per_gpu": wl.n_local * wl.M, "beams": wl.B, "grid": [m.W, m.H], "resolution_m": wl.res, "maps": wl.M, "maps_total": wl.maps_total,
            "parallelism": ("single GPU" if wl.world == 1 else
                            (f"{wl.maps_total} independent maps split over {wl.world} ranks ({wl.M} per GPU), no collective" if wl.batched else
                             f"particles sharded x{wl.world}, map replicated")),
            "exchange": wl.exchange_text(),
            "likelihood_rebuild": "full" if a.full_rebuild else
                                  ("dirty-rect (bit-identical to full)" if os.environ.get("GMS_LIK_SKIP") == "0" else
                                   "dirty-rect; tiles whose thresholded codes the scan does not change are left alone (bit-identical to full; "
                                   "GMS_LIK_SKIP=0 rebuilds every dirty tile); the rebuilds write the scoring factor table only, likelihoodData is "
                                   "materialised on demand outside the timed loop (gms_ensure_lik)"),
            "inputs": "host buffers every step (PCIe-inclusive)" if a.host_inputs else "resident in HBM",
        },
        "value": value,
        "beam_evals_per_s": value * wl.n_hit,
        "scans_per_s": steps * wl.M * (wl.world if wl.batched else 1) / elapsed,
        "kernels": kernels,
        "kernel_symbols": {k: symbols.get(k, k) for k in kernels},
        "filter": {"neff": st0["neff"], "n_zero_weights": st0["n_zero"], "weight_sum": st0["weight_sum"],
                   "note": "the reference's plain product of <= 720 factors: most raw weights underflow at this cloud (reproduced, "
                           "counted); max_log_weight is the underflow-free companion", "max_log_weight": st0["max_log_weight"],
                   "neff_from_log_weights": neff_log},
        "roofline": roof,
        "scan_footprint": {"visited_cells": visits, "dirty_box_cells": dirty_box_cells, "rebuilt_cells_per_step_per_map": dirty,
                           "likelihood_tiles_per_step": None if tiles is None else {k: round(v, 2) for k, v in tiles.items()}},
    }
    return out


def map_update_ms(wl: Workload, nb: int) -> float:
    """BASELINE metric (ii), map-update ms/scan = integrateObservation + computeLikelihoodMap: the map entry point by
    itself (ray cast, apply, likelihood rebuild as three kernels at a fixed device-resident pose), wall time."""
    torch, m, B = wl.torch, wl.m, wl.B
    if wl.batched:
        return None
    upd_pose = torch.from_numpy(np.ascontiguousarray(wl.pose0, dtype=np.float32)).to(wl.dev)

    def one(i):
        bp = wl.beams_ptr(wl.T // 2 + i % wl.n_sets)
        if wl.args.full_rebuild:
            m.integrate_dev(bp, B, upd_pose.data_ptr()); m.compute_likelihood_map()
        else:
            m.update_dev(bp, B, upd_pose.data_ptr())
    for i in range(5):
        one(i)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for i in range(nb):
        one(i)
    torch.cuda.synchronize()
    return (time.perf_counter() - t1) / nb * 1e3


def cpu_baseline(wl: Workload, budget_s: float, with_score_sweep: bool):
    """the oracle (port of the Java loops), one thread, bounded sample of the same workload (map 0 of a batch)."""
    from oracle import oracle as orc
    synth = wl.synth
    ext, res, B, T, n_sets = wl.ext, wl.res, wl.B, wl.T, wl.n_sets
    g = orc.Grid(ext, ext, res, -ext / 2, -ext / 2)
    if wl.batched:
        tr = wl.traces[0]
        log = g.new_log()
        for t in range(T // 2):
            g.integrate(log, tr.scans[t], tr.poses[t])
    else:
        tr = wl.tr
        log = wl.log0.copy()
    lik = g.build_likelihood(log)
    n_cpu = wl.n_local
    P = synth.make_particles(tr.poses[T // 2], n_cpu, seed=99)
    r01 = wl.r01[:, 0]
    # (ii) map update alone: integrateObservation + computeLikelihoodMap (GridMap.java:173-250), full rebuild as the reference does
    reps, u0 = 0, time.perf_counter()
    log_u = log.copy()
    while reps < 3 or (time.perf_counter() - u0 < 0.15 * budget_s and reps < 64):
        g.integrate(log_u, tr.scans[T // 2 + reps % n_sets], tr.poses[T // 2 + reps % n_sets])
        g.build_likelihood(log_u)
        reps += 1
    upd_ms = (time.perf_counter() - u0) / reps * 1e3
    c0 = time.perf_counter()
    done = 0
    for i in range(10 ** 6):
        t = T // 2 + (i % n_sets)
        w = g.score(lik, tr.scans[t], P)
        ws, strongest = orc.normalize(w)
        ne = orc.neff(w) if ws > 0 else float("nan")
        if ws > 0 and ne < n_cpu / 2:
            orc.resample_indices(w, float(r01[i % 4096]))
        wp = orc.weighted_pose(P, w) if ws > 0 else tr.poses[t]
        g.integrate(log, tr.scans[t], wp)
        lik = g.build_likelihood(log)
        done += 1
        if time.perf_counter() - c0 > budget_s:
            break
    cpu_s = time.perf_counter() - c0
    out = {
        "value": n_cpu * done / cpu_s, "unit": "particle-scan evals/s", "cores": 1, "kind": "port",
        "sample": f"{done} scan steps of the same trace" + (" (map 0 of the batch)" if wl.batched else "") + f", all {n_cpu} particles "
                  f"(C oracle, gcc -O2 -ffp-contract=off, single thread; full likelihood rebuild per scan as the reference does)",
        "seconds": cpu_s,
        "map_update_ms_per_scan": upd_ms,
        "map_update_sample": f"{reps} x (integrateObservation + computeLikelihoodMap), C oracle, single thread",
    }
    if not wl.batched and wl.spf is None:
        # the checker's other job here: the device's resampling step against the sequential reference on the SAME normalised
        # weights ("indices equal for fixed r"; the device scans in blocks and flags slots whose threshold lies
        # within rounding distance of a boundary) -- as numbers
        try:
            pf = wl.pf
            pf.set_poses(wl.pose_sets_host[0])
            pf.score_dev(wl.beams_ptr(T // 2), wl.B)
            pf.normalize()
            wn = np.asarray(pf.get_weights(), dtype=np.float64).reshape(-1)
            idx, amb = pf.resample(float(r01[0]), want_indices=True)
            want, _ = orc.resample_indices(wn, float(r01[0]))
            idx = np.asarray(idx, dtype=np.int64).reshape(-1)
            out["resample_agreement"] = {"slots": int(idx.size), "slots_differing": int((idx != want).sum()),
                                         "max_index_distance": int(np.abs(idx - want).max()), "n_ambiguous": int(amb),
                                         "against": "SLAM.resample's running sum (oracle) over the device's normalised weights, same r"}
        except Exception as e:
            out["resample_agreement"] = {"error": repr(e)}
    if with_score_sweep:
        # extra information: scoring alone (a pure function) over host threads with OpenMP; the thread count that does
        # best is reported (containers often expose more CPUs than they may use)
        try:
            navail = len(os.sched_getaffinity(0))
        except AttributeError:
            navail = os.cpu_count() or 1
        w_mt = g.score_mt(lik, tr.scans[T // 2], P, min(navail, 8))
        mt_s, ncore = float("inf"), 1
        th = 2
        while th <= navail:
            m0 = time.perf_counter()
            reps_mt = 0
            while time.perf_counter() - m0 < 0.4:
                g.score_mt(lik, tr.scans[T // 2 + reps_mt % n_sets], P, th)
                reps_mt += 1
            per = (time.perf_counter() - m0) / reps_mt
            if per < mt_s:
                mt_s, ncore = per, th
            th *= 2
        st_s = float("inf")
        for _ in range(3):
            s0 = time.perf_counter()
            w_st = g.score(lik, tr.scans[T // 2], P)
            st_s = min(st_s, time.perf_counter() - s0)
        assert np.array_equal(w_mt, w_st)
        out["score_only"] = {"single_thread_particle_evals_per_s": n_cpu / st_s, "openmp_particle_evals_per_s": n_cpu / mt_s,
                             "openmp_threads": ncore, "cpus_visible": navail,
                             "note": "probabilityOf over all particles only (no map update, no likelihood rebuild); the GPU scoring "
                                     "kernel alone does particles / kernels.score.ms_per_step"}
    return out


# ---------------------------------------------------------------------------------------------------------------------
def trace_replay(path: str, steps: int, warmup: int, particles: int, extent: float, res: float, local_rank: int, torch):
    """A recorded trace (DataRecorder format, gridmap_slam_robot_amd/trace.py) through the device path frame by frame, as
    GridMapApp.onHandleData runs it (J/app/GridMapApp.java:133-192): read_trace -> ONE call per frame, gms_slam_frame (the raw
    measurements handed over as host arrays, 17 bytes per measurement over PCIe; de-skew and motion-model sample in one launch,
    then the fused scan step: five launches).  The first frames only map
    (dead-reckoned pose); `steps` frames are timed, the recording repeated as often as needed (its drive is a closed circle)."""
    from gridmap_slam_robot_amd import GridMap, ParticleFilter, synth
    from gridmap_slam_robot_amd.replay import TraceReplay
    from gridmap_slam_robot_amd.trace import read_trace
    frames = read_trace(path)
    B = max(len(f.angle) for f in frames)
    m = GridMap(extent, extent, res, (-extent / 2, -extent / 2), device=local_rank, max_beams=max(2048, B))
    m.set_stream(torch.cuda.current_stream().cuda_stream)
    pf = ParticleFilter(m, particles)
    # the pose the recording starts from: one odometry step before its first frame's end pose (the synthetic recording's drive)
    start = synth.true_pose(synth.make_world(extent, 4321), -1, len(frames))
    rp = TraceReplay(m, pf, start, seed=2024)
    boot = min(6, len(frames) // 4)
    for f in frames[:boot]:
        rp.bootstrap(f)
    r01 = np.random.default_rng(3).random(4096)
    k = boot
    def nxt():
        nonlocal k
        f = frames[k]
        k = k + 1 if k + 1 < len(frames) else 0
        return f
    for i in range(warmup):
        rp.step(nxt(), float(r01[i % 4096]))
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        rp.step(nxt(), float(r01[(warmup + i) % 4096]))
    issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    st = pf.stats()
    est = pf.weighted_pose()
    out = {"recording": os.path.relpath(path, ROOT), "frames_in_recording": len(frames), "beams": B, "particles": particles,
           "grid": [m.W, m.H], "resolution_m": res, "bootstrap_frames": boot, "frames_timed": steps,
           "ms_per_frame": el / steps * 1e3, "host_issue_ms_per_frame": issue / steps * 1e3, "scans_per_s": steps / el,
           "particle_scan_evals_per_s": particles * steps / el,
           "inputs": "raw polar measurements from host memory every frame (PCIe-inclusive), one gms_slam_frame call per frame; de-skew, motion model, weights, resampling and map on the device",
           "calls_per_frame": ["gms_slam_frame"] if rp.one_call else ["gms_map_deskew", "gms_pf_sample_motion", "gms_slam_update_dev"],
           "launches_per_frame": 5 if rp.one_call else 6,
           "neff_last": st["neff"], "weighted_pose_last": [float(x) for x in est]}
    pf.close(); m.close()
    return out


def particle_maps_run(*args, **kwargs):
    """_particle_maps_run with the collector off: a collection inside one of its few-millisecond regions -- the previous runs' handles
    and tensors going, a hipFree each -- was a third of the region (the default run's first per-particle entry read update_ms 67 us
    where the same build, alone in its process, reads 46)"""
    import gc
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        return _particle_maps_run(*args, **kwargs)
    finally:
        if was:
            gc.enable()


def _particle_maps_run(torch, local_rank: int, particles: int, extent: float, res: float, beams: int, steps: int, warm_frames: int = 12,
                      cpu_seconds: float = 0.0, refine: bool = False):
    """The reference's own filter shape (SLAM.java: one GridMapData per particle; gms_slam_*): `particles` particles with a map of
    extent x extent metres each, scans of `beams` measurements of a synthetic drive through a room that fits the map.  Timed, inputs
    resident in HBM, nothing read back: `steps` SLAM.update calls (motion model inside, no resampling: the maps keep growing), then
    update / resample pairs (SLAM.resample as the filter runs it) and `steps` draws back to back (both arrays moved per call); then
    the pairs with HIP-event brackets around every launch for the per-kernel table.  Rooflines by algorithmic bytes: computeLikelihoodMap 16 B per cell and particle; the resampling copy 32 B per
    cell and particle (both arrays read and written, GridMap.java:118-121)."""
    from gridmap_slam_robot_amd import SLAMParticleMaps, synth
    from gridmap_slam_robot_amd._lib import BEAM_DTYPE
    T = 48
    frames, truth = synth.make_recording(extent, beams, T=T, seed=77)
    start = synth.true_pose(synth.make_world(extent, 77), -1, T)
    dev = torch.device("cuda", local_rank)
    s = SLAMParticleMaps(extent, extent, res, (-extent / 2, -extent / 2), num_particles=particles, device=local_rank, max_beams=max(128, beams))
    s.grid_map.set_stream(torch.cuda.current_stream().cuda_stream)
    s.set_poses(np.tile(np.asarray(start, np.float32), (particles, 1)))
    if refine:
        s.set_refine(True)           # SLAM.java:96: findBestPose of every particle against its own field before it is weighted
    scans, odo, hits = [], [], []
    for f in frames:
        obs = s.grid_map.deskew(f.angle, f.distance, f.hit, f.d_center, f.d_theta)
        hits.append(int(obs.beams["hit"].astype(bool).sum()))
        scans.append(torch.from_numpy(obs.beams.view(np.uint8).reshape(-1).copy()).to(dev))
        odo.append((f.d_center, f.d_theta))
    cells = s.W * s.H
    k = 0
    def step(seq):
        nonlocal k
        s.update_dev(scans[k].data_ptr(), beams, odo[k], seed=11, sequence=seq)
        k = (k + 1) % T
    r01 = np.random.default_rng(5).random(4096)
    pre = (preroll_steps() * 2) // 3                   # untimed, as in measure(): the clocks ramp over the first tens of milliseconds
    for i in range(pre + warm_frames):                 # the maps are explored (the drive has gone round) when the timed region starts
        step(i)
        if i % 4 == 3:
            s.resample(float(r01[i % 4096]))
    torch.cuda.synchronize()
    warm_frames += pre                                  # (sequence numbers go on from here)
    # Every figure below is the median of three regions of `steps` steps that start from the SAME poses (a region is a few milliseconds
    # and one host hiccup inside it -- a collection, a scheduler tick of a busy box -- is a third of it; without the reset the particles,
    # which diffuse while nothing resamples them, would hand every later region another workload).
    regions = {}
    def timed(name, body, reps=3):
        nonlocal k
        P0, k0 = s.get_particles()[0].copy(), k
        out = []
        for _ in range(reps):
            s.set_poses(P0)
            k = k0
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for i in range(steps):
                body(i)
            torch.cuda.synchronize()
            out.append((time.perf_counter() - t0) / steps)
        regions[name] = [v * 1e3 for v in out]
        return sorted(out)[len(out) // 2]
    upd = timed("update_ms", lambda i: step(warm_frames + i))
    st = s.pf.stats()
    # SLAM.resample as the filter runs it: update, resample, update, ... (the pair's time less the update's).  The library copies
    # logData at once and likelihoodData only if somebody reads it before the next update's computeLikelihoodMap has overwritten every
    # cell of it (gms_slam::lazy_lik; GMS_SLAM_LAZY_LIK_COPY=0: both at once) ...
    lazy = os.environ.get("GMS_SLAM_LAZY_LIK_COPY", "1") != "0"
    def pair_body(i):
        step(warm_frames + steps + i)
        s.resample(float(r01[(17 + i) % 4096]))
    pair = timed("update_resample_pair_ms", pair_body)
    rsm = max(pair - upd, 0.0)
    # the reference's loop with its rule decided on the device (gms_slam_resample_maps_if): update + conditional resample, nothing read back
    def rev_body(i):
        step(warm_frames + 3 * steps + i)
        s.resample_if(float(r01[(217 + i) % 4096]), 0.5)
    rev = timed("revolution_ms", rev_body)
    # the same revolution while no resampling step is due (fraction 0: Neff < 0 never holds): the rule is evaluated on the device, nothing
    # is drawn, and no map is copied -- which generation of the maps is current is a device-side fact (gms_slam_resample_maps_if)
    copied0 = s.maps_copied()
    def idle_body(i):
        step(warm_frames + 4 * steps + i)
        s.resample_if(float(r01[(317 + i) % 4096]), 0.0)
    rev_idle = timed("revolution_no_resample_ms", idle_body)
    assert s.maps_copied() == copied0, "a revolution whose rule says no must not copy a map"
    # ... and `steps` draws back to back: every call first brings likelihoodData up to date, so both arrays move per call
    rsm_both = timed("resample_both_arrays_ms", lambda i: s.resample(float(r01[(517 + i) % 4096])))
    # per kernel: event brackets (they cost ~2 us each on the stream: these durations are upper bounds of the un-bracketed ones)
    s.grid_map.profile(True)
    s.grid_map.profile_reset()
    for i in range(steps):
        step(warm_frames + 2 * steps + i)
        s.resample(float(r01[(99 + i) % 4096]))
    torch.cuda.synchronize()
    prof = s.grid_map.profile_get()
    s.grid_map.profile(False)
    kern = {name: {"avg_launch_us": ms / n * 1e3, "launches": n} for name, (ms, n) in prof.items() if n}
    lik_b, copy_b = 16.0 * cells * particles, 32.0 * cells * particles
    moved_b = copy_b / 2 if lazy else copy_b           # per launch of the copy kernel in the loop above
    if "likelihood" in kern:
        kern["likelihood"].update(algorithmic_bytes_per_launch=lik_b, achieved_TBps=lik_b / (kern["likelihood"]["avg_launch_us"] * 1e-6) / 1e12,
                                  hbm_frac=lik_b / (kern["likelihood"]["avg_launch_us"] * 1e-6) / 8e12, what="computeLikelihoodMap of every particle's map (GridMap.java:233-250): 16 B per cell",
                                  us_per_map=kern["likelihood"]["avg_launch_us"] / particles)
    if "mapcopy" in kern:
        kern["mapcopy"].update(algorithmic_bytes_per_launch=moved_b, achieved_TBps=moved_b / (kern["mapcopy"]["avg_launch_us"] * 1e-6) / 1e12,
                               hbm_frac=moved_b / (kern["mapcopy"]["avg_launch_us"] * 1e-6) / 8e12,
                               what=("resample()'s deep copy of logData, map[m] <- map[idx[m]] (SLAM.java:41-45, GridMap.java:120): 16 B per cell moved; likelihoodData's "
                                     "(:121) is deferred and never needed on the path" if lazy else
                                     "resample()'s deep copies, map[m] <- map[idx[m]] for both arrays (SLAM.java:41-45, GridMap.java:118-121): 32 B per cell"))
    if "refine" in kern:
        # GridMap.findBestPose per particle (GridMap.java:319-346): 11 x 11 x 10 lattice poses x the scan's hit beams look-ups of the particle's
        # own field.  Yardsticks: the L1 gather ceiling that binds the shared-map k_refine / k_score_c (profiles/r04/microbench.json: 828 G
        # independent 8-byte look-ups per second) -- this kernel reads its field from LDS instead and runs past it --
        look = 1210.0 * float(np.mean(hits)) * particles
        kern["refine"].update(lookups_per_launch=look, lookups_per_s=look / (kern["refine"]["avg_launch_us"] * 1e-6),
                              l1_gather_ceiling_per_s=828e9, over_l1_gather_ceiling=look / (kern["refine"]["avg_launch_us"] * 1e-6) / 828e9,
                              what="k_slam_refine: findBestPose of every particle against its own field (SLAM.java:96): the field staged in the CU's LDS "
                                   "as probabilityOf's factors where it fits (120 x 120 cells), cell coordinates per (theta, dx | dy, beam) in 16-bit LDS tables, "
                                   "the product per lattice pose in beam order; maps too large for the LDS are read from memory")
    if "score" in kern:
        kern["score"]["what"] = "k_slam_particle: motion sample, probabilityOf against the particle's own field (one lane's product in beam order), integrateObservation into the particle's own map through an LDS count tile"
    # the like-for-like figure: the same updates with every cell of every particle's likelihoodData rebuilt by every update, as the
    # reference does (GMS_SLAM_EAGER_LIK=1, read when a handle is created); the default evaluates the field under the scan's end points
    # only and writes likelihoodData when a caller asks for it -- the same values either way
    eager_ms = None
    on_demand = os.environ.get("GMS_SLAM_EAGER_LIK", "0") != "1" and not refine
    if on_demand:
        old_env = os.environ.get("GMS_SLAM_EAGER_LIK")
        os.environ["GMS_SLAM_EAGER_LIK"] = "1"
        try:
            e = SLAMParticleMaps(extent, extent, res, (-extent / 2, -extent / 2), num_particles=particles, device=local_rank, max_beams=max(128, beams))
        finally:
            if old_env is None:
                os.environ.pop("GMS_SLAM_EAGER_LIK", None)
            else:
                os.environ["GMS_SLAM_EAGER_LIK"] = old_env
        e.grid_map.set_stream(torch.cuda.current_stream().cuda_stream)
        e.set_poses(np.tile(np.asarray(start, np.float32), (particles, 1)))
        for i in range(24):
            e.update_dev(scans[i % T].data_ptr(), beams, odo[i % T], seed=11, sequence=i)
            if i % 4 == 3:
                e.resample(float(r01[i % 4096]))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            e.update_dev(scans[(24 + i) % T].data_ptr(), beams, odo[(24 + i) % T], seed=11, sequence=100 + i)
        torch.cuda.synchronize()
        eager_ms = (time.perf_counter() - t0) / steps * 1e3
        e.close()
    out = {"workload": f"SLAM.java's own shape: {particles} particles x one {s.W}x{s.H} map each @ {res} m, {beams} beams per scan; SLAM.update per particle "
                       "(motion model, computeLikelihoodMap" + (" evaluated under the scan's end points -- likelihoodData itself on demand, the same values; the every-cell rebuild is update_every_cell_rebuilt_ms" if on_demand else "") +
                       ", " + ("findBestPose against the particle's own field, " if refine else "") +
                       "probabilityOf, integrateObservation), SLAM.resample with its deep copies of the maps ("
                       + ("logData at once, likelihoodData when it is read: the next update overwrites it first" if lazy else "both arrays at once") + ")",
           "particles": particles, "grid": [s.W, s.H], "beams": beams, "steps": steps, "refine": bool(refine), "mean_hit_beams": float(np.mean(hits)),
           "update_ms": upd * 1e3, "regions_ms": regions, "updates_per_s": 1.0 / upd, "particle_scan_evals_per_s": particles / upd,
           "likelihood_on_demand": bool(on_demand),
           "update_every_cell_rebuilt_ms": eager_ms,
           "update_every_cell_rebuilt_what": "the same update with computeLikelihoodMap over every cell of every particle's map, as the reference and the CPU port run it "
                                              "(GMS_SLAM_EAGER_LIK=1): the like-for-like figure; the default evaluates the field under the scan's end points only (same values)",
           "resample_ms": rsm * 1e3, "resample_what": ("in the update / resample loop; logData copied, likelihoodData's copy deferred (overwritten by the next "
                                                       "update's computeLikelihoodMap before anything reads it; materialised on demand)" if lazy else
                                                       "in the update / resample loop; both arrays copied at once"),
           "revolution_no_resample_ms": rev_idle * 1e3,
           "revolution_no_resample_what": "the same revolution while no resampling step is due: the rule decided on the device, no draw, no map copied (the reference does nothing either)",
           "revolution_ms": rev * 1e3, "revolution_what": "SLAM.update + `if (neff < n / 2) resample()` (GridMapApp.java:185-186) with the rule decided on the device: no host round trip",
           "resample_both_arrays_ms": rsm_both * 1e3, "resample_copy_TBps_algorithmic": copy_b / rsm_both / 1e12, "resample_copy_hbm_frac": copy_b / rsm_both / 8e12,
           "gridmapdata_bytes_on_device": 4.0 * 8 * cells * particles, "neff_last": st["neff"], "n_zero_weights": st["n_zero"],
           "kernels": kern}
    if cpu_seconds > 0:
        from oracle import oracle as orc                  # the reported CPU baseline of this mode: the oracle's SLAM loop, one thread
        g = orc.Grid(extent, extent, res, -extent / 2, -extent / 2)
        o = orc.Slam(g, particles)
        o.set_poses(np.tile(np.asarray(start, np.float32), (particles, 1)))
        zs = [orc.deskew(f.angle, f.distance, f.hit, f.d_center, f.d_theta) for f in frames]
        t0 = time.perf_counter()
        n = 0
        while n < 2 or (time.perf_counter() - t0 < cpu_seconds and n < T):
            o.update(zs[n % T], odo[n % T], seed=11, sequence=n)
            n += 1
        el = time.perf_counter() - t0
        before = orc.set_threads(1)                       # (the oracle's copies are an OpenMP loop for the tests' sake: one thread here)
        o.resample(0.37)                                  # the first call allocates the second generation's buffers: not timed
        t1 = time.perf_counter()
        o.resample(0.61)
        rs = time.perf_counter() - t1
        orc.set_threads(before)
        out["cpu_baseline"] = {"value": particles / (el / n), "unit": "particle-scan evals/s", "kind": "port", "cores": 1, "seconds": el,
                               "update_ms": el / n * 1e3, "resample_ms": rs * 1e3,
                               "sample": f"{n} SLAM.update calls of the same recording (oracle/gms_oracle.c::orc_slam_update), one orc_slam_resample with its copies on one thread"}
    s.close()
    return out


def particle_maps_sharded_run(torch, dist, rank: int, world: int, local_rank: int, particles: int, extent: float, res: float, beams: int, steps: int):
    """The reference's own filter shape over the ranks of this run: particles WITH their maps, no replica (gms_slam_create_shard +
    distributed.ShardedSlamParticleMaps over torch.distributed = RCCL).  `particles` is the WHOLE population (fixed: strong scaling);
    every rank holds particles / world of them and their maps.  Timed between barriers, maximum over the ranks: `steps` SLAM.update calls
    (the per-particle body + the two small weight collectives), then update / resample pairs with the caller's rule (the draw, the
    all-gather of the sources, the records of the particles that crossed a rank boundary).  This pool has one GPU per box: with more
    than one rank this path has only ever run as two gloo ranks on one device (tests/test_gpu_bench_two_ranks.py); unmeasured on xGMI."""
    from gridmap_slam_robot_amd import synth
    from gridmap_slam_robot_amd._lib import GMS_BLOCK
    from gridmap_slam_robot_amd.distributed import ShardedSlamParticleMaps, SlamShardOps, TorchCollectives
    assert particles % world == 0 and (world == 1 or (particles // world) % GMS_BLOCK == 0), "equal blocks of a multiple of GMS_BLOCK particles"
    n = particles // world
    T = 48
    frames, _ = synth.make_recording(extent, beams, T=T, seed=77)
    start = synth.true_pose(synth.make_world(extent, 77), -1, T)
    ops = SlamShardOps(extent, extent, res, (-extent / 2, -extent / 2), n, rank * n, particles, device=local_rank, max_beams=max(128, beams))
    ops.slam.set_poses(np.tile(np.asarray(start, np.float32), (n, 1)))
    f = ShardedSlamParticleMaps(particles, ops, coll=TorchCollectives())
    scans = [(ops.slam.grid_map.deskew(fr.angle, fr.distance, fr.hit, fr.d_center, fr.d_theta), (fr.d_center, fr.d_theta)) for fr in frames]
    r01 = np.random.default_rng(5).random(4096)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()

    def timed(fn, count):
        barrier()
        t0 = time.perf_counter()
        for i in range(count):
            fn(i)
        barrier()
        el = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([el], dtype=torch.float64, device=ops.device)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        return el / count
    for i in range(24 + (preroll_steps() * 2) // 3):     # untimed: the maps are explored and the clocks have ramped when the timed regions start
        f.update(scans[i % T][0], scans[i % T][1], seed=11, sequence=i)
        if i % 4 == 3:
            f.resample(float(r01[i % 4096]))
    upd = timed(lambda i: f.update(scans[(24 + i) % T][0], scans[(24 + i) % T][1], seed=11, sequence=100 + i), steps)
    sent0, res0 = f.records_sent, f.resamples

    def pair(i):
        f.update(scans[(24 + i) % T][0], scans[(24 + i) % T][1], seed=11, sequence=1000 + i)
        f.resample(float(r01[100 + i]), 0.5)
    pr = timed(pair, steps)
    nres = f.resamples - res0
    moved = torch.tensor([float(f.records_sent - sent0)], dtype=torch.float64, device=ops.device)
    if world > 1:
        dist.all_reduce(moved)
    rec_bytes = ops.record_doubles * 8
    out = {"workload": f"SLAM.java's own shape sharded: {particles} particles x one {ops.slam.W}x{ops.slam.H} map each @ {res} m over {world} rank(s) "
                       f"({n} particles and their maps per rank, no replica), {beams} beams per scan; SLAM.update = per-particle body + block partials "
                       "all-reduced + packed particles all-gathered; SLAM.resample = the draw per rank + the records of the particles that crossed a rank boundary",
           "particles": particles, "particles_per_rank": n, "ranks": world, "grid": [ops.slam.W, ops.slam.H], "beams": beams, "steps": steps,
           "update_ms": upd * 1e3, "particle_scan_evals_per_s": particles / upd, "update_resample_pair_ms": pr * 1e3,
           "resample_ms": max(pr - upd, 0.0) * 1e3, "resampling_steps_timed": nres,
           "records_moved_per_resample": (float(moved.item()) / nres) if nres else 0.0, "record_bytes": rec_bytes,
           "bytes_moved_per_resample": (float(moved.item()) / nres * rec_bytes) if nres else 0.0,
           "neff_last": f.stats()["neff"],
           "measured_on": "one GPU per box in this pool: figures with more than one rank come from ranks sharing a device over gloo, never from xGMI"}
    ops.slam.close()
    return out


def explore_run(args, torch, local_rank: int, skip: bool, passes: int = 5):
    """Config 3 on a map that is being EXPLORED: the map starts empty and the timed steps are the drive's first T/2 scans, so every
    scan pushes the frontier on and the tiles along it change their thresholded codes (the headline step runs on a pre-built map
    that is revisited, where the unchanged-tile rule leaves nearly every dirty tile alone).  `passes` passes of T/2 scan steps are
    timed, the map reset (untimed) in front of each; one more pass runs with the tile census on (gms_map_tile_stats) and one with
    the event brackets on.  skip = False: the same with GMS_LIK_SKIP=0, every dirty tile rebuilt.  The reference rebuilds the
    whole field on every scan (J/slam/GridMap.java:233-250)."""
    from gridmap_slam_robot_amd import GridMap, ParticleFilter, synth
    cfg = synth.CONFIGS["C3"]
    n, B, ext, res = cfg["particles"], cfg["beams"], cfg["extent"], cfg["resolution"]
    T = 64
    half = T // 2
    dev = torch.device("cuda", local_rank)
    tr = synth.make_trace(ext, res, B, T=T, seed=1234)
    old = os.environ.get("GMS_LIK_SKIP")
    os.environ["GMS_LIK_SKIP"] = "1" if skip else "0"           # read when the handle is created
    try:
        m = GridMap(ext, ext, res, (-ext / 2, -ext / 2), device=local_rank, max_beams=max(2048, B))
    finally:
        if old is None:
            os.environ.pop("GMS_LIK_SKIP", None)
        else:
            os.environ["GMS_LIK_SKIP"] = old
    m.set_stream(torch.cuda.current_stream().cuda_stream)
    pf = ParticleFilter(m, n)
    scans = torch.from_numpy(tr.scans.view(np.uint8).reshape(T, -1).copy()).to(dev)
    beam_ptrs = [scans[t].data_ptr() for t in range(T)]
    sets = [torch.from_numpy(np.ascontiguousarray(synth.make_particles(tr.poses[t], n, seed=99 + t))).to(dev) for t in range(half)]
    pose_ptrs = [p.data_ptr() for p in sets]
    true_poses = torch.from_numpy(np.ascontiguousarray(tr.poses[:half], dtype=np.float32)).to(dev)
    r01 = np.random.default_rng(11).random(half)

    def fresh():
        m.reset()
        m.compute_likelihood_map()
        torch.cuda.synchronize()

    def scan_steps():
        for t in range(half):
            pf.slam_update_dev(pose_ptrs[t], beam_ptrs[t], B, float(r01[t]), 0.5, True)

    def map_updates():
        for t in range(half):
            m.update_dev(beam_ptrs[t], B, true_poses[t].data_ptr())

    out = {}
    for name, body in (("step", scan_steps), ("map_update", map_updates)):
        fresh(); body(); torch.cuda.synchronize()               # warm-up pass
        tot = 0.0
        for _ in range(passes):
            fresh()
            t0 = time.perf_counter()
            body()
            torch.cuda.synchronize()
            tot += time.perf_counter() - t0
        out[name + "_ms"] = tot / (passes * half) * 1e3
        fresh()
        m.tile_stats(True, fetch=False)
        body()
        ts = m.tile_stats(False)
        out[name + "_tiles_per_scan"] = {k: round(v / half, 2) for k, v in ts.items()}
    fresh()
    m.profile(True); m.profile_reset()
    scan_steps()
    torch.cuda.synchronize()
    prof = m.profile_get()
    m.profile(False)
    st = pf.stats()
    res_ = {"workload": f"C3 on an EMPTY map, the drive's first {half} scans ({n} particles x {B} beams, {m.W}x{m.H} @ {res} m): the frontier moves "
                        "every scan" + ("" if skip else "; GMS_LIK_SKIP=0: every dirty tile rebuilt"),
            "steps_timed": passes * half, "ms_per_step": out["step_ms"], "value": n / (out["step_ms"] * 1e-3), "unit": "particle-scan evals/s",
            "map_update_ms_per_scan": out["map_update_ms"],
            "tiles_per_scan": {"scan_step": out["step_tiles_per_scan"], "map_update": out["map_update_tiles_per_scan"],
                               "legend": "64x32-cell tiles of the dirty box per scan: left alone (no code changes) / uniform and already holding "
                                         "its constants / uniform, constants written / blurred (both passes)"},
            "bracketed_us_per_launch": {KERNEL_SYMBOLS.get(k, k): round(ms / nn * 1e3, 2) for k, (ms, nn) in prof.items() if nn},
            "neff_last": st["neff"]}
    pf.close(); m.close()
    return res_


# ---------------------------------------------------------------------------------------------------------------------
# the ONE stdout line: compact (a few KB), strict JSON; everything else goes to the report file it names
# ---------------------------------------------------------------------------------------------------------------------
KERNEL_SYMBOLS = {"score": "k_score_c", "reduce": "k_partials", "raycast": "k_norm_raycast", "likelihood": "k_lik_resample",
                  "apply": "k_apply", "resample": "k_resample", "order": "k_order", "exchange": "ncclAllGather (grouped)"}
LINE_LIMIT = 4096       # bytes; the driver reads the line from a bounded tail of stdout (round 3's 22 KB line was cut: parsed = null)


def _sig(x, digits=6):
    """numbers to `digits` significant figures (the report file keeps them in full)"""
    if isinstance(x, bool) or x is None or isinstance(x, (int, str)):
        return x
    if isinstance(x, float):
        if not math.isfinite(x):
            return None                      # strict JSON has no NaN / Infinity
        return float(f"{x:.{digits}g}")
    if isinstance(x, dict):
        return {k: _sig(v, digits) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_sig(v, digits) for v in x]
    return x


def compact_line(full: dict, report_file: str | None) -> str:
    """The contract's fields and the few figures a reader needs beside them, as one JSON line below LINE_LIMIT bytes.
    `full` is the complete report (what round 3 printed); it is written to `report_file`, which the line names."""
    cfg = full.get("config") or {}
    line = {k: full.get(k) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                                     "scaling", "vs_baseline", "dtype", "data")}
    line["config"] = {k: cfg[k] for k in ("workload", "particles_total", "particles_per_gpu", "beams", "grid", "resolution_m", "maps", "maps_total",
                                          "parallelism", "exchange", "inputs") if cfg.get(k) is not None}
    for k in ("timed_region_s", "timed_regions", "region_ms_per_step", "first_region_ms_per_step", "warmup_effective", "preroll_steps", "map_update_ms_per_scan", "map_update_ms_per_scan_exploring", "beam_evals_per_s", "scans_per_s", "per_rank_ms_per_step", "exchange_latency_us",
              "sharded_equals_standalone", "rccl_ranks"):
        if full.get(k) is not None:
            line[k] = full[k]
    r = full.get("roofline")
    if r:
        rl = {k: r.get(k) for k in ("kernel", "kernel_symbol", "bound", "achieved", "peak", "unit", "frac", "traffic",
                                    "algorithmic_bytes_per_launch", "avg_launch_us", "launches_timed")}
        for k in ("lookup_ceiling_frac", "lookups_per_s", "lookup_ceiling_per_s", "traffic_over_algorithmic"):
            if r.get(k) is not None:
                rl[k] = r[k]
        st = r.get("step") or {}
        rl["step"] = {k: st.get(k) for k in ("frac", "achieved", "algorithmic_bytes_per_step", "launches_per_step", "kernel_us_per_step")}
        line["roofline"] = rl
    else:
        line["roofline"] = None
    cb = full.get("cpu_baseline")
    if cb:
        line["cpu_baseline"] = {k: cb.get(k) for k in ("value", "unit", "cores", "kind", "sample", "seconds", "map_update_ms_per_scan",
                                                       "gpu_like_for_like_ms_per_step")}
    else:
        line["cpu_baseline"] = None
    kern = full.get("kernels") or {}
    if kern:
        sym = full.get("kernel_symbols") or KERNEL_SYMBOLS
        line["kernel_us"] = {sym.get(k, KERNEL_SYMBOLS.get(k, k)): v.get("avg_launch_us") for k, v in kern.items()}
    sec = full.get("secondary") or {}
    if sec:
        # one number per secondary run (ms per step / per frame); the runs themselves are in the report file
        line["secondary_ms_per_step"] = {k: (v.get("ms_per_step", v.get("ms_per_frame")) if "error" not in v else None) for k, v in sec.items()}
    if full.get("trace_replay"):
        line["trace_replay"] = {k: full["trace_replay"].get(k) for k in ("recording", "frames_timed", "ms_per_frame", "particles", "beams")}
    line["report"] = report_file
    text = json.dumps(_sig(line), allow_nan=False, separators=(", ", ": "))
    if len(text) > LINE_LIMIT:              # never print a line the driver cannot take: shed the optional blocks, longest first
        for k in ("secondary_ms_per_step", "kernel_us", "per_rank_ms_per_step", "trace_replay"):
            line.pop(k, None)
            text = json.dumps(_sig(line), allow_nan=False, separators=(", ", ": "))
            if len(text) <= LINE_LIMIT:
                break
    return text


def _json_safe(x):
    """the report file is strict JSON too: NaN / Infinity -> null, numpy scalars -> Python numbers"""
    if isinstance(x, dict):
        return {str(k): _json_safe(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_json_safe(v) for v in x]
    if isinstance(x, (np.floating, float)):
        x = float(x)
        return x if math.isfinite(x) else None
    if isinstance(x, np.integer):
        return int(x)
    if isinstance(x, np.bool_):
        return bool(x)
    if isinstance(x, np.ndarray):
        return _json_safe(x.tolist())
    return x


def emit(full: dict, result_fd: int, report_path: str):
    """writes the full report to `report_path` and the compact line to the saved stdout"""
    full = _json_safe(full)
    name = None
    try:
        os.makedirs(os.path.dirname(os.path.abspath(report_path)), exist_ok=True)
        with open(report_path, "w") as f:
            json.dump(full, f, indent=1, allow_nan=False)
            f.write("\n")
        name = os.path.relpath(os.path.abspath(report_path), ROOT)
    except OSError as e:                     # a read-only tree must not cost the line
        print(f"bench.py: could not write {report_path}: {e}", file=sys.stderr)
    os.write(result_fd, (compact_line(full, name) + "\n").encode())


def launch_command(n_gpus: int, argv, port: int):
    """the command line that runs this file on n_gpus ranks of one node (what the driver's launcher line is)"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_gpus}", "--master-addr", "127.0.0.1",
            "--master-port", str(port), os.path.abspath(__file__)] + list(argv)


def self_launch(n_gpus: int, argv) -> int:
    """Runs the ranks as a child process (never an exec: this process may not have touched the GPU, the child does) and forwards its
    stdout -- rank 0's one JSON line -- and exit code.  GMS_BENCH_LAUNCH_DRYRUN=1 prints the command instead (tests)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = launch_command(n_gpus, argv, port)
    if os.environ.get("GMS_BENCH_LAUNCH_DRYRUN", "") not in ("", "0"):
        print(json.dumps({"launch": cmd}))
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "1")
    sys.stdout.flush()
    return subprocess.run(cmd, env=env).returncode


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--config", default=None,
                    help="C3 (default at 1 GPU: 16384 particles per GPU, weak scaling) | C4 (default at N > 1 GPUs: BASELINE's FIXED population of "
                         "65536 particles split over the ranks, strong scaling) | C2 | C5 (64 batched maps per rank)")
    ap.add_argument("--particles", type=int, default=0, help="override particles per GPU and per map (C4: the whole population)")
    ap.add_argument("--maps", type=int, default=0, help="batched configurations: maps per handle instead of the configuration's (e.g. --config C5 --maps 8: one GPU's share of config 5 on an 8-GPU node)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-secondary", action="store_true", help="skip the C5 / C2 block of the default 1-GPU run")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline's scan-step loop")
    ap.add_argument("--full-rebuild", action="store_true", help="rebuild the whole likelihood field every scan")
    ap.add_argument("--host-inputs", action="store_true", help="hand poses and scans over as HOST buffers every step (PCIe-inclusive rate; never the headline value)")
    ap.add_argument("--exchange", default="auto", choices=["auto", "in-library", "torch-single", "torch-two"],
                    help="sharded runs: first exchange route to try (auto = in-library RCCL; later routes are fall-backs)")
    ap.add_argument("--force-sharded", action="store_true", help="run the sharded (all-gather) code path even with one rank")
    ap.add_argument("--soak-seconds", type=float, default=6.0,
                    help="default 1-GPU line: seconds of closed-loop steps back to back reported as secondary.soak (0 skips it)")
    ap.add_argument("--no-verify", action="store_true", help="sharded runs: skip the sharded == stand-alone check")
    ap.add_argument("--trace", default="", help="replay a recorded trace (DataRecorder format) frame by frame instead of the synthetic C3 step; "
                    "the JSON line then carries the replay under 'trace_replay' and value = particles x frames / s")
    ap.add_argument("--trace-particles", type=int, default=1024)
    ap.add_argument("--trace-extent", type=float, default=25.6, help="map extent (m) for --trace")
    ap.add_argument("--trace-res", type=float, default=0.05, help="map resolution (m) for --trace")
    ap.add_argument("--particle-maps", default="", help="run only the per-particle-map mode (SLAM.java's own shape): PARTICLES,EXTENT_M,BEAMS e.g. 500,6,90")
    ap.add_argument("--refine", action="store_true", help="with --particle-maps: SLAM.update refines every particle's pose (findBestPose against its own field, SLAM.java:96)")
    ap.add_argument("--report", default=os.path.join(ROOT, "bench_report.json"),
                    help="where the full report goes (kernel table, secondary runs, notes); stdout carries one compact line that names it")
    args = ap.parse_args()
    if args.config not in (None, "C2", "C3", "C4", "C5"):
        print("bench.py: --config must be C2, C3, C4 or C5", file=sys.stderr)
        return 2

    # `python bench.py --gpus N` without a launcher: start the N ranks ourselves, exactly as the driver's command line does
    # (python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same arguments>), as a CHILD process and before
    # this process has touched the GPU (nothing above imports torch or loads the library): rank 0's single JSON line reaches our
    # stdout through the child's, and we leave with the child's exit code.
    if args.particle_maps and args.refine and (args.gpus > 1 or int(os.environ.get("WORLD_SIZE", "1")) > 1):
        print("bench.py: --particle-maps --refine is a one-GPU run; use --gpus 1", file=sys.stderr)
        return 2
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return self_launch(args.gpus, sys.argv[1:])

    # stdout carries the ONE JSON line and nothing else: RCCL prints a version banner on C stdout when a
    # communicator is created, so everything but the result goes to stderr
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and args.gpus > 1:
        print(f"bench.py: --gpus {args.gpus} but the launcher started {world} rank(s)", file=sys.stderr)
        return 2
    # Test hooks (tests/test_gpu_bench_two_ranks.py): several ranks on ONE GPU with gloo collectives, to exercise the
    # N > 1 control flow (route fall-back, self-verification, per-rank timing) where no multi-GPU node exists.  RCCL
    # refuses two ranks on one device, so such a run takes the torch.distributed route by construction.
    backend = os.environ.get("GMS_BENCH_DIST_BACKEND", "nccl")
    if os.environ.get("GMS_BENCH_SHARE_DEVICE", "") not in ("", "0"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1 or (args.force_sharded and "RANK" in os.environ):
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    if args.particle_maps and (world > 1 or args.force_sharded):
        # the reference-shape filter sharded over the ranks: particles with their maps (fixed population: strong scaling)
        n_, ext_, b_ = args.particle_maps.split(",")
        pm = particle_maps_sharded_run(torch, dist, rank, world, local_rank, int(n_), float(ext_), 0.05, int(b_), args.steps)
        if rank == 0:
            out = {"metric": "particle-scan evals/sec", "value": pm["particle_scan_evals_per_s"], "unit": "particle-scan evals/s", "n_gpus": world,
                   "steps": args.steps, "warmup": args.warmup, "ms_per_step": pm["update_ms"], "higher_is_better": True, "scaling": "strong",
                   "vs_baseline": None, "dtype": "f64", "data": "synthetic", "config": {"workload": pm["workload"], "parallelism": f"particles with their maps over {world} rank(s)"},
                   "per_particle_maps_sharded": pm, "roofline": None, "cpu_baseline": None}
            emit(out, result_fd, args.report)
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        return 0
    if args.particle_maps:
        n_, ext_, b_ = args.particle_maps.split(",")
        pm = particle_maps_run(torch, local_rank, int(n_), float(ext_), 0.05, int(b_), args.steps, cpu_seconds=0.0 if (args.no_cpu_baseline or args.refine) else 5.0,
                               refine=args.refine)
        out = {"metric": "particle-scan evals/sec", "value": pm["particle_scan_evals_per_s"], "unit": "particle-scan evals/s", "n_gpus": 1,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": pm["update_ms"], "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic", "config": {"workload": pm["workload"]},
               "per_particle_maps": pm, "roofline": None, "cpu_baseline": pm.get("cpu_baseline")}
        lk = pm["kernels"].get("likelihood")
        if lk and "achieved_TBps" in lk:
            # this mode's streaming kernel (the particle kernel is bound by vector issue and by the touched lines' read-modify-write, not by
            # a byte model); duration from the HIP-event brackets (an upper bound: ~2 us of markers), traffic: profiles/r05/per_particle_maps_kernels.json
            out["roofline"] = {"kernel": "likelihood", "kernel_symbol": "k_slam_likelihood", "bound": "hbm", "achieved": lk["achieved_TBps"] * 1e3,
                               "peak": 8000.0, "unit": "GB/s", "frac": lk["hbm_frac"], "traffic": None,
                               "algorithmic_bytes_per_launch": lk["algorithmic_bytes_per_launch"], "avg_launch_us": lk["avg_launch_us"]}
        mc = pm["kernels"].get("mapcopy")
        if out["roofline"] is None and mc and "achieved_TBps" in mc:
            # with likelihoodData on demand no kernel of update() streams: the mode's HBM-bound launch is resample()'s copy of the maps (the
            # particle kernel is bound by vector issue and by the touched lines' read-modify-write, not by a byte model)
            out["roofline"] = {"kernel": "mapcopy", "kernel_symbol": "k_slam_gather_one", "bound": "hbm", "achieved": mc["achieved_TBps"] * 1e3,
                               "peak": 8000.0, "unit": "GB/s", "frac": mc["hbm_frac"], "traffic": None,
                               "algorithmic_bytes_per_launch": mc["algorithmic_bytes_per_launch"], "avg_launch_us": mc["avg_launch_us"]}
        emit(out, result_fd, args.report)
        return 0

    if args.trace:
        tr = trace_replay(args.trace, args.steps, args.warmup, args.trace_particles, args.trace_extent, args.trace_res, local_rank, torch)
        out = {"metric": "particle-scan evals/sec", "value": tr["particle_scan_evals_per_s"], "unit": "particle-scan evals/s", "n_gpus": 1,
               "steps": args.steps, "warmup": args.warmup, "ms_per_step": tr["ms_per_frame"], "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f64", "data": "synthetic recording (tools/make_recording.py)",
               "config": {"workload": f"trace replay: {tr['particles']} particles x {tr['beams']} beams, {tr['grid'][0]}x{tr['grid'][1]} grid @ {tr['resolution_m']} m, "
                                      "de-skew + motion model + full scan step per recorded frame"},
               "trace_replay": tr, "roofline": None, "cpu_baseline": None}
        emit(out, result_fd, args.report)
        return 0

    config = args.config or ("C3" if world == 1 else "C4")      # BASELINE.json configs[2] at one GPU, configs[3] over several
    sharded = world > 1 or args.force_sharded
    want_cpu = world == 1 and not args.no_cpu_baseline

    def run_sharded_or_single(name, steps, warmup, comm=None, keep_log=False):
        """one configuration on this run's ranks: route, self-verification, measurement; rank 0 gets the report"""
        w = Workload(name, args, torch, dist, rank, world, local_rank, sharded, keep_log=keep_log, comm=comm, n_maps=(args.maps or None))
        ver = None
        if w.spf is not None:
            w.pick_route()
            if not args.no_verify:
                try:
                    ver = w.verify_against_standalone()
                except Exception as e:      # the check must never take the measurement down with it
                    print(f"bench.py: sharded-vs-standalone check failed to run: {e!r}", file=sys.stderr)
                    ver = {"sharded_equals_standalone": None, "error": repr(e)} if rank == 0 else None
        ms = measure(w, steps, warmup)
        if rank != 0:
            return w, None
        r = report(w, ms, steps, warmup)
        r["steps"], r["warmup"] = steps, warmup
        r["scaling"] = "strong" if w.strong else "weak"
        mu_ = map_update_ms(w, ms["nb"])
        if mu_ is not None:
            r["map_update_ms_per_scan"] = mu_
        if world > 1 or w.spf is not None:
            r["per_rank_ms_per_step"] = [round(t / steps * 1e3, 5) for t in ms["per_rank"]]
            ex = r["kernels"].get("exchange")
            r["exchange_latency_us"] = ex["avg_launch_us"] if ex else None     # the L of DESIGN.md section 7 (event-bracketed, in-library route)
            if ver is not None:
                r["verify"] = ver
                r["sharded_equals_standalone"] = ver.get("sharded_equals_standalone")
                r["rccl_ranks"] = ver.get("rccl_ranks")
        return w, r

    wl, rep = run_sharded_or_single(config, args.steps, args.warmup, keep_log=(want_cpu and rank == 0))
    # N > 1 with the default configuration: the fixed population is the line; the weak series (16384 particles per GPU, the
    # 1-GPU line's shape on every rank) rides along as secondary.weak -- every rank takes part in both
    weak_rep = None
    if world > 1 and args.config is None and not args.no_secondary and not args.particles:
        try:
            comm = wl.comm
            wl.comm = None                   # (the communicator outlives the first workload: one per device)
            wl.pf.close(); wl.m.close()
            w2, weak_rep = run_sharded_or_single("C3", max(10, min(args.steps, 100)), max(2, min(args.warmup, 10)), comm=comm)
            w2.pf.close(); w2.m.close()
        except Exception as e:
            print(f"bench.py: weak-scaling companion run failed: {e!r}", file=sys.stderr)
            weak_rep = {"error": repr(e)} if rank == 0 else None
    if rank != 0:
        if dist.is_initialized():
            dist.barrier()
            dist.destroy_process_group()
        return 0

    out = {
        "metric": "particle-scan evals/sec",
        "value": rep.pop("value"),
        "unit": "particle-scan evals/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": rep.pop("ms_per_step"),
        "higher_is_better": True,
        "scaling": rep.pop("scaling"),
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
    }
    out.update(rep)
    if weak_rep is not None:
        out["secondary"] = {"weak": weak_rep}

    if want_cpu:
        out["cpu_baseline"] = cpu_baseline(wl, args.cpu_seconds, with_score_sweep=True)
    else:
        out["cpu_baseline"] = None
    # The like-for-like pair of the CPU baseline: the C port rebuilds all cells of the likelihood field on every scan (as the
    # reference does, GridMap.java:233-250); the headline step rebuilds the dirty tiles only.  The same step with every cell
    # rebuilt and likelihoodData written (--full-rebuild's path) is timed here and stated beside the baseline.
    if want_cpu and out["cpu_baseline"] and config == "C3" and not (args.full_rebuild or args.host_inputs or args.force_sharded):
        try:
            fr_args = argparse.Namespace(**vars(args))
            fr_args.full_rebuild = True
            wf = Workload("C3", fr_args, torch, dist, 0, 1, local_rank, False)
            mf = measure(wf, 60, 8)
            out["cpu_baseline"]["gpu_like_for_like_ms_per_step"] = mf["steady"] * 1e3
            out["cpu_baseline"]["gpu_like_for_like_what"] = ("the same scan step through the separate entry points with computeLikelihoodMap over ALL cells and "
                                                             "likelihoodData written every scan (bench.py --full-rebuild), as the CPU port does")
            wf.pf.close(); wf.m.close()
            del wf
        except Exception as e:
            out["cpu_baseline"]["gpu_like_for_like_ms_per_step"] = None
            print(f"bench.py: like-for-like (full rebuild) run failed: {e!r}", file=sys.stderr)

    # ---- the other single-GPU configurations, shorter runs of the same measurement ------------------------------------
    if world == 1 and not args.no_secondary and config == "C3" and not (args.full_rebuild or args.host_inputs or args.particles or args.force_sharded):
        sec = {}
        # C5 and C2 as BASELINE states them; C4_1gpu = the N = 1 point of the fixed-population series (65536 particles on one GPU);
        # C3x8_batched = eight maps of the headline shape in one batched handle (config 5's mode at 2 cm: what the chip does when the
        # four-launch latency chain of a single map is amortised over eight)
        for name, cfg_name, n_maps, (k_steps, k_warm) in (("C5", "C5", None, (20, 3)), ("C2", "C2", None, (100, 10)),
                                                          ("C4_1gpu", "C4", None, (50, 5)), ("C3x8_batched", "C3", 8, (20, 3))):
            try:
                w2 = Workload(cfg_name, args, torch, dist, 0, 1, local_rank, False, keep_log=(want_cpu and n_maps is None and cfg_name != "C4"),
                              n_maps=n_maps)
                m2 = measure(w2, k_steps, k_warm)
                r2 = report(w2, m2, k_steps, k_warm)
                r2["steps"], r2["warmup"] = k_steps, k_warm
                r2["scaling"] = "strong" if w2.strong else "weak"
                mu2 = map_update_ms(w2, m2["nb"])
                if mu2 is not None:
                    r2["map_update_ms_per_scan"] = mu2
                if want_cpu and n_maps is None and cfg_name != "C4":
                    r2["cpu_baseline"] = cpu_baseline(w2, 3.0, with_score_sweep=False)
                sec[name] = r2
                w2.pf.close(); w2.m.close()
                del w2
            except Exception as e:
                sec[name] = {"error": repr(e)}
        # the headline configuration on a map that is being explored (the timed region above revisits a pre-built map)
        for name, skip in (("C3_explore", True), ("C3_explore_every_tile", False)):
            try:
                sec[name] = explore_run(args, torch, local_rank, skip)
            except Exception as e:
                sec[name] = {"error": repr(e)}
        if "error" not in sec["C3_explore"]:
            out["map_update_ms_per_scan_exploring"] = sec["C3_explore"]["map_update_ms_per_scan"]
        # the same configuration as a CLOSED LOOP: motion model and resampling on the device, the particles never leave it; once
        # with the caller's order (what the headline runs), once with the locality order forced on before every scoring launch
        for name, order in (("C3_loop", None), ("C3_loop_ordered", "1")):
            try:
                old_env = os.environ.get("GMS_SCORE_ORDER")
                if order is not None:
                    os.environ["GMS_SCORE_ORDER"] = order
                try:
                    w3 = Workload("C3", args, torch, dist, 0, 1, local_rank, False, loop=True)
                finally:
                    if order is not None:
                        if old_env is None:
                            os.environ.pop("GMS_SCORE_ORDER", None)
                        else:
                            os.environ["GMS_SCORE_ORDER"] = old_env
                m3 = measure(w3, 100, 10)
                r3 = report(w3, m3, 100, 10)
                r3["steps"], r3["warmup"] = 100, 10
                r3["config"]["workload"] += "; closed loop: poses = the previous step's particles moved by gms_pf_sample_motion" + (
                    ", locality order (k_order) before every scoring launch" if order else "")
                for k in ("filter",):
                    r3.pop(k, None)
                sec[name] = r3
                w3.pf.close(); w3.m.close()
                del w3
            except Exception as e:
                sec[name] = {"error": repr(e)}
        # a soak: the closed loop for several seconds on end -- long enough for any utilisation sampler to see the GPU busy, and
        # long enough to show that the filter still knows where the robot is after ~10^5 scans on one map; then the same loop, shorter,
        # with the opt-in log-normalisation (gms_pf_set_log_normalize): how many particles stay alive and where the filter ends up
        for name, seconds, log_norm in (("soak", args.soak_seconds, False), ("soak_log_normalize", min(args.soak_seconds, 2.0), True)):
            if seconds <= 0:
                continue
            try:
                w4 = Workload("C3", args, torch, dist, 0, 1, local_rank, False, loop=True)
                if log_norm:
                    w4.pf.set_log_normalize(True)
                for i in range(10):
                    w4.step(i)
                w4.barrier()
                t0 = time.perf_counter()
                i = 10
                neffs = []
                while time.perf_counter() - t0 < seconds:
                    for _ in range(2000):                          # ~0.1 s of queued work between two waits
                        w4.step(i)
                        i += 1
                    w4.barrier()
                    st = w4.pf.stats()                             # (outside the steps' own work: one read-back per 2000 steps)
                    neffs.append((st[0] if isinstance(st, list) else st)["neff"])
                el = time.perf_counter() - t0
                est = np.asarray(w4.pf.weighted_pose(), dtype=np.float64).reshape(-1)[:3]
                truth = w4.tr.poses[(w4.T // 2 + i - 1) % w4.T].astype(np.float64)
                st = w4.pf.stats()
                st = st[0] if isinstance(st, list) else st
                sec[name] = {"workload": "C3 closed loop (secondary.C3_loop), back to back" + (", log-normalisation on (not the reference's arithmetic)" if log_norm else ""),
                             "seconds": el, "steps": i - 10,
                             "ms_per_step": el / (i - 10) * 1e3, "value": w4.n_local * (i - 10) / el, "unit": "particle-scan evals/s",
                             "pose_error_m": float(math.hypot(est[0] - truth[0], est[1] - truth[1])),
                             "neff": st["neff"], "neff_median_of_samples": float(np.median(neffs)) if neffs else None,
                             "n_zero_weights": st["n_zero"], "weight_sum_finite": bool(np.isfinite(st["weight_sum"]))}
                w4.pf.close(); w4.m.close()
                del w4
            except Exception as e:
                sec[name] = {"error": repr(e)}
        # the reference's own filter shape (SLAM.java: one GridMapData per particle) at its operating point and at a size that
        # no longer fits the caches
        for name, (n_, ext_, b_, k_) in (("per_particle_maps", (500, 6.0, 90, 50)), ("per_particle_maps_refine", (500, 6.0, 90, 50)),
                                         ("per_particle_maps_4096x256", (4096, 12.8, 180, 10))):
            try:
                rf = name.endswith("_refine")
                pm = particle_maps_run(torch, local_rank, n_, ext_, 0.05, b_, k_, cpu_seconds=(3.0 if (want_cpu and n_ <= 1000 and not rf) else 0.0), refine=rf)
                pm["ms_per_step"] = pm["update_ms"]
                sec[name] = pm
            except Exception as e:
                sec[name] = {"error": repr(e)}
        rec = os.path.join(ROOT, "tests", "golden", "recording_360.bin")
        if os.path.exists(rec):
            try:
                sec["trace_replay"] = trace_replay(rec, 200, 20, 1024, 25.6, 0.05, local_rank, torch)
            except Exception as e:
                sec["trace_replay"] = {"error": repr(e)}
        out.setdefault("secondary", {}).update(sec)

    emit(out, result_fd, args.report)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()
    return 0


if __name__ == "__main__":
    sys.exit(main())
