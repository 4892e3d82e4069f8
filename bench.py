#!/usr/bin/env python
"""bench.py -- VB EM iterations/s of the MI355X VBx hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

With N > 1 and no torchrun environment the script re-executes itself under ``python -m torch.distributed.run
--nnodes=1 --nproc-per-node N --master-addr 127.0.0.1`` (one rank per GPU over RCCL); launched by torchrun it reads
RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment.  Every rank checks that RCCL sees N ranks.

Workload (config.workload): every GPU holds a batch of ``--batch`` independent synthetic recordings of the headline
shape T=10 000 x-vectors, R=128, S=30 (vbx_amd.synth, kappa=0.05, random gamma init).  The default of 64 per GPU is
BASELINE.json config 4's batch of 64 recordings resident on one MI355X (it needs < 1 GB of the 288 GB); under weak
scaling every further GPU holds another 64.  One *step* = one VB EM iteration (M-step, log-likelihoods,
forward-backward, ELBO, pi update: VBx.py:94-105) of every recording in the batch.  ``value`` counts
recording-iterations per second over all ranks; inputs are resident in HBM before the timed region.  Weak scaling:
per-GPU work is fixed, recordings never talk to each other, RCCL carries only the barrier / max-over-ranks of the
timings.

Timing: W untimed warm-up steps, then blocks of EXACTLY K steps, each bracketed by barrier + synchronize on both
sides and reduced with MAX over the ranks; blocks are repeated until the timed region covers ``--min-seconds``
(0.5 s by default: one block of the driver's 20 steps is 7 ms, too short to be stable) and the MEDIAN block gives
``ms_per_step`` and ``value``; the spread is reported beside it.

Also on the same JSON line:
  roofline                  dominant kernel: algorithmic bytes / HIP-event time vs the 8 TB/s HBM peak
  roofline_whole_iteration  the five launches of an iteration together, against the survey's byte count (SURVEY 8d:
                            one kernel per stage) and against the fused design's compulsory bytes
  f32_exact                 the headline (`value`) multiplies rho alpha^T and gamma^T rho with f16 operand pairs on the matrix
                            cores (VBX_OPT_GEMM = split: v_mfma_f32_16x16x32_f16, three products per pair, fp32 storage and
                            accumulation; vbx_amd/csrc/vbx_split.hpp) -- the headline since round 5 under the condition the
                            round-4 review set: every parity entry of tests/test_gpu_configs.py within north_star's 1e-4 (all
                            nine C5 points against the extended-precision referee included), geometric mean of the deviations
                            split / exact <= 1.2, a recording's result independent of its batch (tests/test_gpu_split.py).
                            f32_exact = the same batch with the exact f32 matrix instruction (--gemm exact makes it the
                            headline and the split the sub-record f32_split), with its own roofline
  f64                       the same batch on the fp64 path (what vbhmm.py gets: its inputs are float64), with its own roofline
  configs                   BASELINE.json configs[1], [2], [4]: C2 (T=10k, S=10), C3 (T=50k, S=30), C5 (T=200k, S=50, loopProb
                            0.9, the nine-point Fa/Fb sweep as one batch on a shared rho, and with private copies), fp32 and
                            fp64: ms per iteration, dominant kernel, algorithmic bytes, fraction of the HBM peak
  single_recording          latency-bound rate of ONE recording (batch=1) on one GPU
  cpu_baseline              the NumPy/SciPy restatement oracle/vbx_oracle.py (kind "port": the reference's algorithm and
                            third-party calls, pinned to the reference's outputs at this very size by
                            tests/test_oracle_golden.py) on the host cores, bounded sample (rank 0, N=1 only).  Always the
                            port: the same thing is timed on every box, and nothing of the reference travels

Output: the FULL record goes to ``--full-out`` (default gpurun_out/bench_full.json) and to stderr; the LAST line of
stdout is a compact JSON (< 4 KB, compact_record()) with the contract's keys, ``roofline``, ``cpu_baseline`` and a digest of
the sub-records -- what the driver parses.
"""
from __future__ import annotations

import argparse
import json
import os
os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')   # before any HIP runtime starts (torch included): the library's stream
                                                    # groups want one hardware queue per stream (vbx_host_state.hpp)
import socket
import statistics
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0           # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec

# algorithmic HBM bytes per recording per launch, as (passes over T x R, passes over T x S); fp32 storage = 4 bytes,
# fp64 = 8.  SURVEY.md 8d: 8*T*R + 28*T*S per iteration for one kernel per stage; the fused kernels need fewer passes.
ALGO_PASSES = {
    'mstep_acc': (1, 1),      # rho read, gamma read
    'loglik': (1, 1),         # rho read, b write
    'fb': (0, 3),             # b read by forward and by backward, ahat write
    'post': (0, 2),           # ahat read, gamma write
    'chunk_loglik': (1, 1),   # fused loglik + chunk operator: rho read, b write
    'chunk_post': (1, 1),     # fused re-run + posteriors + next gamma^T rho: rho read, b read (gamma stays on the chip)
}


# A kernel CLASS of the library's event timers (vbx_batch_kernel_times) may be several kernels in a profile: the forward-backward
# classes of the paths that do not run the fused per-chunk kernels (64 < S: vbx_scan_wide.hpp; S > 256: the sequential walk)
CLASS_MEMBERS = {'fb': ('scan1_wide', 'scan3_wide', 'scan1', 'scan3', 'fb_seq', 'fb_big'),
                 'fb_aux': ('scan2_wide', 'scan2', 'scan_compose')}


def _members(kernel, table):
    """the entries of a profile's kernel table that make up ``kernel`` (itself, or the members of its class)"""
    if kernel in table:
        return [kernel]
    return [m for m in CLASS_MEMBERS.get(kernel, ()) if m in table]


def pmc_traffic(kernel, workload):
    """HBM bytes per launch of ``kernel`` from the committed rocprofv3 PMC passes (tools/pmc_traffic.py) ->
    (bytes, file, document), or (None, reason, None).  A profile counts only if it was taken on exactly this workload AND
    on the kernels this tree builds: the file carries the hash of the kernel sources it was measured on
    (vbx_amd.build.iteration_source_hash), and a file whose hash differs is refused -- a stale figure is worse than none."""
    import glob
    from vbx_amd.build import iteration_source_hash
    now = iteration_source_hash()
    best, stale = None, None
    for path in sorted(glob.glob(os.path.join(REPO, 'profiles', '*_pmc_traffic.json'))):
        try:
            doc = json.load(open(path))
        except (OSError, ValueError):
            continue
        members = _members(kernel, doc.get('kernels', {}))
        if doc.get('workload') != workload or not members:
            continue
        if doc.get('iteration_source_sha16') != now:
            stale = os.path.relpath(path, REPO)
            continue
        best = (sum(doc['kernels'][m]['hbm_bytes_per_launch'] for m in members), os.path.relpath(path, REPO), doc)
    if best:
        return best
    return (None, f'refused: {stale} was measured on other kernel sources than {now}' if stale else 'no PMC profile of this workload on file', None)


def sq_issue(kernel, workload):
    """Issue-side picture of ``kernel`` from the committed SQ pass (tools/pmc_counters.py -> profiles/*_sq_issue.json):
    fractions of all SIMD cycles of a launch in which a matrix instruction executes (SQ_VALU_MFMA_BUSY_CYCLES) and in which
    a vector instruction issues (SQ_ACTIVE_INST_VALU x 4 cycles), and what is left.  Same rules as pmc_traffic: the file
    must be of this workload and of these kernel sources.  -> dict or None"""
    import glob
    from vbx_amd.build import iteration_source_hash
    now = iteration_source_hash()
    for path in sorted(glob.glob(os.path.join(REPO, 'profiles', '*_sq_issue.json'))):
        try:
            doc = json.load(open(path))
        except (OSError, ValueError):
            continue
        members = _members(kernel, doc.get('kernels', {}))
        if doc.get('workload') != workload or not members or doc.get('iteration_source_sha16') != now:
            continue
        ks = [doc['kernels'][m] for m in members]                  # (a class of several kernels: weighted by their durations)
        w = [k.get('duration_us_under_profiler', 1.0) for k in ks]
        mfma = sum(k['mfma_busy'] * x for k, x in zip(ks, w)) / sum(w)
        valu = sum(k['valu_issue'] * x for k, x in zip(ks, w)) / sum(w)
        return {'mfma_busy': mfma, 'valu_issue': valu, 'idle': max(0.0, 1.0 - mfma - valu), 'source': os.path.relpath(path, REPO),
                'kernels': members}
    return None


HBM_COPY_GBS = 6290.0           # MI355X_MICROARCH.md: 6.29 TB/s measured with a float4 copy (79 % of the spec)


def classify_bound(issue, traffic_bytes, avg_us):
    """What a kernel is bound by, from the counters and not from the design's intention:
      'simd-issue'  the SIMDs issue matrix or vector instructions in >= 70 % of their cycles
      'hbm'         the bytes it really moves (PMC) go at >= 70 % of what a plain copy reaches on this chip
      'latency'     neither: exposed waits (dependent chains, launches too small to fill the chip)
      'unprofiled'  no SQ / PMC pass of this workload on these kernel sources is on file"""
    if issue is None and traffic_bytes is None:
        return 'unprofiled'
    if issue is not None and issue['mfma_busy'] + issue['valu_issue'] >= 0.70:
        return 'simd-issue'
    if traffic_bytes is not None and traffic_bytes / (avg_us * 1e-6) / 1e9 >= 0.70 * HBM_COPY_GBS:
        return 'hbm'
    return 'latency'


def algo_bytes(kernel, T, R, S, esize, n_rec=1, rho_copies=None):
    """Algorithmic HBM bytes of one launch over ``n_rec`` recordings of which ``rho_copies`` (default: all) have a rho of
    their own -- a sweep over one recording reads ONE rho (vbx_batch_set_recording_shared)."""
    pr, ps = ALGO_PASSES[kernel]
    return esize * (pr * T * R * (n_rec if rho_copies is None else rho_copies) + ps * T * S * n_rec)


def make_batch(ctx, n_rec, T, S, D, precision, seed0, max_iters, streams=None):
    from vbx_amd import _capi
    from vbx_amd.synth import make_recording
    batch = _capi.Batch(ctx, [T] * n_rec, [S] * n_rec, D, precision=precision, max_iters=max_iters)
    if streams is not None and batch.streams != streams:
        batch.set_option(_capi.OPT_STREAMS, streams)
    for b in range(n_rec):
        X, Phi, _ = make_recording(T, S, D=D, seed=seed0 + b, kappa=0.05, dtype=np.float32)
        g = np.random.default_rng(10_000 + seed0 + b).gamma(1.0, size=(T, S))
        g /= g.sum(1, keepdims=True)
        batch.set_recording(b, X, Phi, np.ones(S) / S, g, 0.99, 0.3, 17.0)
    return batch


SWEEP_POINTS = [(fa, fb) for fa in (0.2, 0.3, 0.4) for fb in (6.0, 17.0, 64.0)]     # DIHARD2_run.sh:45-46, AMI_run.sh:47, CALLHOME_run.sh:45-46


def make_sweep_batch(ctx, T, S, D, precision, max_iters, shared, loop_prob=0.9, seed=0, streams=None):
    """BASELINE configs[4]: ONE recording under the nine (Fa, Fb) points of the recipes' grids, every point from the same
    random initialisation.  ``shared``: the points read one rho (vbx_batch_set_recording_shared); otherwise every point is
    a recording of its own with a private copy of the same x-vectors (what round 2 ran)."""
    from vbx_amd import _capi
    from vbx_amd.synth import make_recording
    from vbx_amd.batch import sweep_streams
    n = len(SWEEP_POINTS)
    want = streams if streams is not None else (sweep_streams(n, T) if shared else 1)
    batch = _capi.Batch(ctx, [T] * n, [S] * n, D, precision=precision, max_iters=max_iters, streams=want)
    assert batch.streams == want
    X, Phi, _ = make_recording(T, S, D=D, seed=seed, kappa=0.05, dtype=np.float32)
    g = np.random.default_rng(10_000 + seed).gamma(1.0, size=(T, S)).astype(np.float32)
    g /= g.sum(1, keepdims=True)
    for k, (fa, fb) in enumerate(SWEEP_POINTS):
        if shared and k > 0:
            batch.set_recording_shared(k, 0, np.ones(S) / S, g, loop_prob, fa, fb)
        else:
            batch.set_recording(k, X, Phi, np.ones(S) / S, g, loop_prob, fa, fb)
    return batch


def cpu_baseline(T, S, D, iters, precision='fp32', parity_iters=8, reference_path=None):
    """The oracle (kind "port": oracle/vbx_oracle.py -- the reference's algorithm, arithmetic order and third-party calls
    [scipy.special.logsumexp per frame], pinned to the reference's outputs at this very size by
    tests/test_oracle_golden.py) timed on one core.  The same recording also serves BASELINE.json's second metric, max |gamma -
    gamma_NumPy| of the GPU path -- as the MAXIMUM over iterations 1 ... parity_iters (round 6: the reference's contract is "any
    maxIters", VBx.py:91, and the deviation of an fp32 path peaks at iterations 2-3 of a random start; the oracle shows its
    trajectory as a chain of one-iteration calls, which is the loop's own arithmetic: its state is (gamma, pi), VBx.py:87-104).
    ``reference_path`` (--cpu-reference, opt-in): a checkout of the reference; its own VBx/VBx.py is timed in a subprocess on
    the same inputs (kind "reference") -- nothing of it is copied or imported into this process."""
    import contextlib
    import io
    from oracle import vbx_oracle                              # the checker / the baseline, never the product path
    from vbx_amd.synth import make_recording
    X, Phi, _ = make_recording(T, S, D=D, seed=0, kappa=0.05)
    g = np.random.default_rng(10_000).gamma(1.0, size=(T, S))
    g /= g.sum(1, keepdims=True)
    kw = dict(loopProb=0.99, Fa=0.3, Fb=17.0, epsilon=-1e300)
    t0 = time.perf_counter()
    with contextlib.redirect_stdout(io.StringIO()):
        vbx_oracle.VBx(X, Phi, maxIters=iters, pi=S, gamma=g, **kw)
    dt = time.perf_counter() - t0
    out = {'value': iters / dt, 'unit': 'EM iterations/s', 'cores': 1, 'kind': 'port',
           'sample': f'{iters} iterations of one recording T={T} S={S} R={D} (oracle/vbx_oracle.py, float64, NumPy + SciPy '
                     f'logsumexp per frame as VBx.py:167-171, {dt:.1f} s on {os.cpu_count()} visible cores; the path is single-threaded)'}
    import vbx_amd
    worst = {'gamma_max_abs_diff': 0.0, 'pi_max_abs_diff': 0.0, 'elbo_max_rel_diff': 0.0, 'at_iteration': 0}
    per_iteration = []
    g_ref, pi_ref = g, np.ones(S) / S
    with contextlib.redirect_stdout(io.StringIO()):
        for k in range(1, parity_iters + 1):
            g_ref, pi_ref, L_ref = vbx_oracle.VBx(X, Phi, maxIters=1, pi=pi_ref, gamma=g_ref, **kw)
            g_gpu, pi_gpu, L_gpu = vbx_amd.VBx(X, Phi, maxIters=k, pi=S, gamma=g, precision=precision, **kw)   # a fresh call of k iterations
            dg = float(np.abs(g_gpu - g_ref).max())
            per_iteration.append(dg)
            if dg >= worst['gamma_max_abs_diff']:
                worst['gamma_max_abs_diff'], worst['at_iteration'] = dg, k
            worst['pi_max_abs_diff'] = max(worst['pi_max_abs_diff'], float(np.abs(pi_gpu - pi_ref).max()))
            worst['elbo_max_rel_diff'] = max(worst['elbo_max_rel_diff'], abs(L_gpu[-1][0] - L_ref[0][0]) / abs(L_ref[0][0]))
    out['parity_over_iterations'] = dict(worst, iterations=parity_iters, gamma_max_abs_diff_per_iteration=per_iteration,
                                         precision=precision, target=1e-4,
                                         note='max over iterations 1 ... n, each a fresh GPU call with maxIters = k against the oracle trajectory')
    if reference_path:
        out['reference'] = cpu_reference(reference_path, T, S, D, iters)
    return out


def cpu_reference(path, T, S, D, iters):
    """The UNMODIFIED reference (``<path>/VBx/VBx.py::VBx``) timed in a subprocess on the inputs of cpu_baseline ("Run
    /root/reference/VBx/VBx.py::VBx unmodified").  Opt-in (--cpu-reference PATH): the GPU box has no copy of the reference, and
    nothing of it is ever copied into this repository."""
    import subprocess
    src = os.path.join(path, 'VBx', 'VBx.py')
    if not os.path.exists(src):
        return {'error': f'{src} not found'}
    code = (
        'import importlib.util, io, contextlib, json, sys, time\n'
        'import numpy as np\n'
        f'sys.path.insert(0, {REPO!r})\n'
        'from vbx_amd.synth import make_recording\n'
        f'spec = importlib.util.spec_from_file_location("_ref_VBx", {src!r}); mod = importlib.util.module_from_spec(spec); spec.loader.exec_module(mod)\n'
        f'T, S, D, iters = {T}, {S}, {D}, {iters}\n'
        'X, Phi, _ = make_recording(T, S, D=D, seed=0, kappa=0.05)\n'
        'g = np.random.default_rng(10_000).gamma(1.0, size=(T, S)); g /= g.sum(1, keepdims=True)\n'
        't0 = time.perf_counter()\n'
        'with contextlib.redirect_stdout(io.StringIO()):\n'
        '    out = mod.VBx(X, Phi, loopProb=0.99, Fa=0.3, Fb=17.0, pi=S, gamma=g, maxIters=iters, epsilon=-1e300)\n'
        'dt = time.perf_counter() - t0\n'
        'print(json.dumps({"seconds": dt, "iterations": len(out[2]), "elbo_last": float(out[2][-1][0])}))\n')
    res = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True)
    if res.returncode != 0:
        return {'error': res.stderr[-400:]}
    d = json.loads(res.stdout.strip().splitlines()[-1])
    return {'value': d['iterations'] / d['seconds'], 'unit': 'EM iterations/s', 'cores': 1, 'kind': 'reference',
            'sample': f'{d["iterations"]} iterations of one recording T={T} S={S} R={D}: {src} unmodified, in a subprocess, {d["seconds"]:.1f} s',
            'elbo_last': d['elbo_last']}


def call_level(n_rec, T, S, D, precision, iters=40, reps=3):
    """The headline batch as ONE reference-style batch call with host arrays in and out (vbx_amd.batch.VBx_batch: allocation,
    H2D of n x (X, gamma0), the iterations, the gamma write-out, D2H of n x gamma) -- the PCIe-inclusive rate SURVEY 8d asks
    for beside the HBM-resident `value`; never `value` itself."""
    from vbx_amd.batch import VBx_batch
    from vbx_amd.synth import make_recording
    recs = []
    for b in range(n_rec):
        X, Phi, _ = make_recording(T, S, D=D, seed=b, kappa=0.05, dtype=np.float32)
        g = np.random.default_rng(10_000 + b).gamma(1.0, size=(T, S)).astype(np.float32)
        g /= g.sum(1, keepdims=True)
        recs.append(dict(X=X, Phi=Phi, pi=S, gamma=g))
    kw = dict(maxIters=iters, epsilon=-1e300, loopProb=0.99, Fa=0.3, Fb=17.0, precision=precision)
    VBx_batch(recs, **kw)                                      # (first call: pinned result blocks, device blocks, library warm-up)
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        out = VBx_batch(recs, **kw)
        ts.append(time.perf_counter() - t0)
        del out
    dt = statistics.median(ts)
    return {'value': n_rec * iters / dt, 'unit': 'recording-EM-iterations/s (call level: host arrays in and out)', 'ms_per_call': 1e3 * dt,
            'ms_per_call_min': 1e3 * min(ts), 'iterations': iters, 'recordings': n_rec, 'precision': precision, 'calls_timed': reps,
            'note': 'one vbx_amd.batch.VBx_batch call: uploads enqueued from one host thread per stream (ABI 7), one synchronize, '
                    'results into pinned host memory; float32 inputs, float64 gamma out'}


COMPACT_LIMIT = 4096            # bytes: the driver parses the LAST stdout line; round 4's 25 KB line came back unparsed


def _sig(x, n=6):
    """Numbers to n significant digits (the compact line is for reading and parsing, the full record keeps everything)."""
    if isinstance(x, bool) or not isinstance(x, float):
        return x
    if x != x or x in (float('inf'), float('-inf')):
        return None
    return float(f'{x:.{n}g}')


def compact_record(out):
    """The last stdout line: the contract's keys verbatim, ``roofline`` and ``cpu_baseline`` reduced to their numeric
    fields, and one digest entry {value, ms, frac, bound_today} per sub-record.  Pure function of the full record (tested on CPU,
    tests/test_host_and_abi.py); guaranteed < COMPACT_LIMIT bytes -- digest entries are dropped from the end if ever needed."""
    keep = ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling',
            'vs_baseline', 'dtype', 'gemm', 'data')
    c = {k: _sig(out[k], 9) for k in keep if k in out}
    cfg = out.get('config', {})
    c['config'] = {k: cfg[k] for k in ('workload', 'recordings_per_gpu', 'T', 'R', 'S', 'streams_per_gpu', 'parallelism') if k in cfg}
    if len(c['config'].get('workload', '')) > 200:
        c['config']['workload'] = c['config']['workload'][:197] + '...'
    r = out.get('roofline')
    if r:
        c['roofline'] = {k: _sig(r.get(k)) for k in ('bound', 'kernel', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'traffic_source',
                                                        'algorithmic_bytes_per_launch', 'avg_launch_us')}
        c['roofline']['bound_today'] = r.get('bound_today')
    w = out.get('roofline_whole_iteration')
    if w:
        c['step_frac_of_hbm_peak'] = _sig(w.get('fused_compulsory_frac_of_hbm_peak'))
    cb = out.get('cpu_baseline')
    if cb:
        c['cpu_baseline'] = {k: _sig(cb.get(k)) for k in ('value', 'unit', 'cores', 'kind')}
        c['cpu_baseline']['sample'] = cb.get('sample', '')[:160]
        par = cb.get('parity_over_iterations')
        if par:
            c['gamma_max_abs_diff_vs_numpy'] = _sig(par['gamma_max_abs_diff'], 3)
            c['gamma_max_abs_diff_over_iterations'] = f"1..{par['iterations']} (max at {par['at_iteration']})"
        if cb.get('reference') and 'value' in cb['reference']:
            c['cpu_baseline_reference'] = {k: _sig(cb['reference'].get(k)) for k in ('value', 'unit', 'cores', 'kind')}

    def digest(d):
        e = {'value': _sig(d.get('value'), 5), 'ms': _sig(d.get('ms_per_step', d.get('ms_per_iteration', d.get('ms_per_call'))), 5)}
        roof = d.get('roofline') or d
        if roof.get('frac') is not None:
            e['frac'] = _sig(roof['frac'], 3)
            e['bound_today'] = roof.get('bound_today', roof.get('bound'))   # (what the counters say limits it; every frac is against the HBM peak)
        return e
    subs = {}
    for key in ('f32_exact', 'f32_split', 'f64', 'single_recording', 'strong_scaling_form', 'call_level'):
        if key in out:
            subs[key] = digest(out[key])
    for key, d in out.get('configs', {}).items():
        subs[key] = digest(d)
    c['configs'] = subs
    c['full_record'] = out.get('full_record')
    line = json.dumps(c, separators=(',', ':'))
    names = list(subs)
    while len(line) >= COMPACT_LIMIT and names:                # (never needed at today's 25 sub-records: ~3 KB)
        del subs[names.pop()]
        c['configs_truncated'] = True
        line = json.dumps(c, separators=(',', ':'))
    assert len(line) < COMPACT_LIMIT, len(line)
    return line


def emit(out, full_path):
    """Full record -> file (+ stderr); compact line -> the last line of stdout."""
    if full_path:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(full_path)), exist_ok=True)
            with open(full_path, 'w') as fh:
                json.dump(out, fh, indent=1)
            out['full_record'] = os.path.relpath(os.path.abspath(full_path), REPO)
        except OSError as e:
            out['full_record'] = f'not written: {e}'
    print(json.dumps(out), file=sys.stderr, flush=True)
    print(compact_record(out), flush=True)


def respawn_under_torchrun(n):
    """`python bench.py --gpus N` with N > 1 and no rendezvous environment: become N ranks."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={n}',
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--batch', type=int, default=64, help='recordings per GPU')
    ap.add_argument('--T', type=int, default=10000)
    ap.add_argument('--S', type=int, default=30)
    ap.add_argument('--D', type=int, default=128)
    ap.add_argument('--precision', default='fp32', choices=['fp32', 'fp64', 'fp32-split'],
                    help="fp32-split = --precision fp32 --gemm split (the name the library and the profiles use)")
    ap.add_argument('--gemm', default='split', choices=['exact', 'split'],
                    help='how the fp32 HEADLINE multiplies: split (default since round 5) = f16 operand pairs on the matrix cores, '
                         'f32 accumulation (VBX_OPT_GEMM = split, precision="fp32-split"); exact = v_mfma_f32_16x16x4_f32.  The '
                         'other one is measured as a sub-record either way')
    ap.add_argument('--no-split', action='store_true', help='skip the sub-record of the other GEMM mode')
    ap.add_argument('--total-recordings', type=int, default=None,
                    help='strong scaling: this many recordings over ALL ranks (BASELINE configs[3] as stated: 64 over 8 GPUs = 8 '
                         'per GPU); --batch is then total / gpus and "scaling" says strong')
    ap.add_argument('--min-seconds', type=float, default=0.5, help='timed region: blocks of K steps until this much time')
    ap.add_argument('--max-blocks', type=int, default=200)
    ap.add_argument('--cpu-iters', type=int, default=20, help='oracle iterations for cpu_baseline (0 = skip)')
    ap.add_argument('--parity-iters', type=int, default=8, help='iterations 1 ... n over which gamma_max_abs_diff_vs_numpy is the maximum')
    ap.add_argument('--cpu-reference', default=None, metavar='PATH',
                    help='a checkout of the reference (e.g. /root/reference): also time its own VBx/VBx.py, unmodified, in a subprocess '
                         '(cpu_baseline.reference, kind "reference"); opt-in, never copied')
    ap.add_argument('--no-call-level', action='store_true', help='skip the call-level (host arrays in and out) record of the headline batch')
    ap.add_argument('--no-single', action='store_true', help='skip the batch=1 latency measurement')
    ap.add_argument('--no-f64', action='store_true', help='skip the fp64 sub-record')
    ap.add_argument('--no-configs', action='store_true', help='skip the C2 / C3 / C5 records (BASELINE.json configs[1,2,4])')
    ap.add_argument('--streams', type=int, default=None, help='HIP streams per batch (default: the library\'s choice)')
    ap.add_argument('--full-out', default=os.path.join(REPO, 'gpurun_out', 'bench_full.json'),
                    help="where the full record goes ('' = nowhere); stdout carries the compact line only")
    ap.add_argument('--dry-run', action='store_true', help='stop after the ranks are established (no GPU needed)')
    args = ap.parse_args()
    if args.precision == 'fp32-split':
        args.precision, args.gemm = 'fp32', 'split'

    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        respawn_under_torchrun(args.gpus)              # (does not return)
    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'--gpus {args.gpus} but WORLD_SIZE={world}')
    if args.dry_run:
        print(json.dumps({'rank': rank, 'local_rank': local_rank, 'world': world}), flush=True)
        return

    import torch
    # One GPU per rank over RCCL.  (Test hook, tests/test_gpu_multirank.py: VBX_AMD_DIST_BACKEND=gloo VBX_AMD_DEVICE=0 runs
    # the same N-rank code path with every rank on the one GPU of a test box -- RCCL refuses two ranks on one device; such a
    # line says so in `config.parallelism` and is not a scaling measurement.)
    backend = os.environ.get('VBX_AMD_DIST_BACKEND', 'nccl')
    device = int(os.environ.get('VBX_AMD_DEVICE', local_rank)) if backend != 'nccl' else local_rank
    cdev = 'cuda' if backend == 'nccl' else 'cpu'      # where the few scalars of the timing protocol live
    torch.cuda.set_device(device)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if backend == 'nccl':
            dist.init_process_group(backend='nccl', device_id=torch.device('cuda', device))
        else:
            dist.init_process_group(backend=backend)
        ones = torch.ones(1, device=cdev)
        dist.all_reduce(ones)                          # RCCL really connects N ranks, one per GPU
        assert dist.get_world_size() == args.gpus and int(ones.item()) == args.gpus, 'RCCL does not see --gpus ranks'
        if backend == 'nccl':
            assert torch.cuda.device_count() >= args.gpus, f'{torch.cuda.device_count()} GPUs visible, --gpus {args.gpus}'

    from vbx_amd import _capi
    ctx = _capi.Context(device)
    info = ctx.device_info()
    esize = 4 if args.precision == 'fp32' else 8
    K, W = args.steps, args.warmup
    if args.total_recordings is not None:
        if args.total_recordings % world:
            raise SystemExit(f'--total-recordings {args.total_recordings} is not a multiple of {world} ranks')
        args.batch = args.total_recordings // world
    # the precision string the library takes for the headline and for the other GEMM mode of the fp32 path
    head_prec = 'fp32-split' if (args.precision == 'fp32' and args.gemm == 'split') else args.precision
    other_prec = None if args.precision != 'fp32' else ('fp32' if args.gemm == 'split' else 'fp32-split')

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def timed_blocks(batch, min_seconds, max_blocks):
        """Blocks of exactly K steps, each between barrier + synchronize, MAX over the ranks; every rank runs the same
        number of blocks (rank 0 decides)."""
        times = []
        while True:
            barrier()
            t0 = time.perf_counter()
            batch.run(K, -np.inf)                      # returns after the streams have drained
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            barrier()
            if dist is not None:
                t = torch.tensor([dt], dtype=torch.float64, device=cdev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            times.append(dt)
            go = torch.tensor([1 if (sum(times) < min_seconds and len(times) < max_blocks) else 0], device=cdev)
            if dist is not None:
                dist.broadcast(go, src=0)
            if not int(go.item()):
                return times

    def kernel_probe(make, min_seconds, max_blocks):
        """Kernel-level figures of a workload on ONE stream: with several streams per GPU (the library's default for a
        big batch) launches of different streams share the CUs, and the duration of a launch says how it shared them, not
        what the kernel achieves.  HIP events on the batch's own stream: 8 untimed iterations with events around every
        launch (which kernels run at all?), then W + blocks of K iterations with events around the launches of the
        HBM-side kernels only.  -> ({kernel: {avg_us, launches}}, dominant HBM-side kernel, ms per step, blocks)"""
        probe = make(W + K * min(max_blocks, 64) + 32, 1)
        probe.profile_kernels(None)
        probe.run(8, -np.inf)
        survey = probe.kernel_times()
        per_kernel = {k: {'avg_us': 1e3 * ms / n, 'launches': n} for k, (ms, n) in survey.items() if n}
        # (the survey runs on cold clocks; the HBM-side kernels of an iteration are close to each other, so all of them
        #  are bracketed over the warm steps and the dominant one is chosen from those averages)
        cands = [k for k in per_kernel if k in ALGO_PASSES and per_kernel[k]['launches'] >= 8]   # (not the one-off first accumulation)
        probe.profile_kernels(cands)
        probe.run(W, -np.inf)
        acc = {k: [0.0, 0] for k in cands}
        probe_ms, blocks, t_probe = 0.0, 0, time.perf_counter()
        while blocks < max(1, min(max_blocks, (probe.max_iters - W - 16) // K)) and \
                (blocks == 0 or time.perf_counter() - t_probe < min_seconds):
            probe.run(K, -np.inf)                      # (like the timed region: blocks of exactly K steps, repeated to a
            kt = probe.kernel_times()                  #  minimum duration -- one block of the driver's 20 steps is 6 ms)
            for k in cands:
                acc[k][0] += kt[k][0]
                acc[k][1] += kt[k][1]
            probe_ms += probe.last_run_ms()[0]
            blocks += 1
        for k in cands:
            per_kernel[k] = {'avg_us': 1e3 * acc[k][0] / acc[k][1], 'launches': acc[k][1]}
        dom = max(cands, key=lambda k: per_kernel[k]['avg_us'])
        probe.close()
        return per_kernel, dom, probe_ms / (K * blocks), blocks

    def roofline_of(per_kernel, dom, n_rec, T, S, D, es, workload, rho_copies=None):
        """``roofline`` of the dominant HBM-side kernel: algorithmic bytes of one launch / its HIP-event duration, the bytes it
        really moved (PMC) and what its SIMDs did meanwhile (SQ) where a profile of this workload on these kernel sources is
        on file.  ``bound`` = "hbm": the roofline the path is priced against (the algorithm is on the bandwidth side, DESIGN
        section 4) -- ``achieved`` / ``frac`` are always against the HBM peak; ``bound_today`` = what the counters say limits
        the kernel as it stands (classify_bound)."""
        dom_bytes = algo_bytes(dom, T, D, S, es, n_rec, rho_copies)
        avg_us = per_kernel[dom]['avg_us']
        achieved = dom_bytes / (avg_us * 1e-6) / 1e9
        traffic = pmc_traffic(dom, workload)
        issue = sq_issue(dom, workload)
        out = {'bound': 'hbm', 'bound_today': classify_bound(issue, traffic[0], avg_us), 'kernel': dom, 'achieved': achieved, 'peak': HBM_PEAK_GBS,
               'unit': 'GB/s', 'frac': achieved / HBM_PEAK_GBS, 'traffic': traffic[0], 'traffic_source': traffic[1],
               'algorithmic_bytes_per_launch': dom_bytes, 'avg_launch_us': avg_us, 'simd_issue': issue}
        if traffic[0] is not None:
            out['traffic_GBs'] = traffic[0] / (avg_us * 1e-6) / 1e9
            out['traffic_frac_of_copy_rate'] = out['traffic_GBs'] / HBM_COPY_GBS
        return out, traffic

    # upper bound on the iterations a batch will be asked for (the ELBO history lives on the device)
    budget = W + K * (args.max_blocks + 2) + 16

    def headline(precision, max_iters, streams, n_rec=None):
        n_rec = args.batch if n_rec is None else n_rec
        return make_batch(ctx, n_rec, args.T, args.S, args.D, precision, seed0=rank * n_rec,
                          max_iters=max_iters, streams=streams)

    per_kernel, dom, probe_ms_per_step, probe_blocks = kernel_probe(lambda mi, st: headline(head_prec, mi, st),
                                                                     args.min_seconds / 2, args.max_blocks)

    # ---- the timed region: the library's default configuration, no per-kernel events
    batch = headline(head_prec, budget, args.streams)
    streams = batch.streams
    batch.run(W, -np.inf)
    times = timed_blocks(batch, args.min_seconds, args.max_blocks)
    res0 = batch.result(0, want_model=False)
    gemm_ran = batch.gemm
    batch.close()
    med = statistics.median(times)
    if args.precision == 'fp32':
        assert gemm_ran == args.gemm, f'--gemm {args.gemm} but the library multiplied {gemm_ran}'

    # ---- a multi-GPU run also measures BASELINE configs[3] AS STATED -- 64 recordings over ALL ranks, 64 / N per GPU
    # (strong scaling) -- next to the weak-scaling headline, so that one driver sweep N = 1, 2, 4, 8 yields both curves
    strong = None
    STATED_TOTAL = 64
    if world > 1 and args.total_recordings is None and STATED_TOTAL % world == 0:
        bs = headline(head_prec, budget, args.streams, n_rec=STATED_TOTAL // world)
        bs.run(W, -np.inf)
        ts_ = timed_blocks(bs, args.min_seconds / 2, args.max_blocks)
        strong = (ts_, bs.streams)
        bs.close()

    # ---- the other configurations of BASELINE.json, each with its own kernel-level figure (rank 0 measures what fits one
    # GPU by itself; the multi-GPU metric is the batch above).  value = recording-iterations/s like the headline.
    def one_config(name, make, n_rec, T, S, precision, note, min_seconds, rho_copies=None):
        es = 8 if precision == 'fp64' else 4
        pk, dk, _, _ = kernel_probe(make, min_seconds / 2, 16)
        b = make(W + K * 18 + 16, None)
        b.run(W, -np.inf)
        ts = []
        while sum(ts) < min_seconds and len(ts) < 16:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            b.run(K, -np.inf)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        elbo = float(b.result(0, want_gamma=False, want_model=False)['Li'][-1])
        nstreams = b.streams
        b.close()
        dt = statistics.median(ts)
        workload = {'batch': n_rec, 'T': T, 'S': S, 'D': args.D, 'precision': precision}
        if name.startswith('C5'):
            workload['sweep'] = name.split('_')[-1]
        roof, _ = roofline_of(pk, dk, n_rec, T, S, args.D, es, workload, rho_copies)
        fused = ((n_rec if rho_copies is None else rho_copies) * 8 * T * args.D + n_rec * 8 * T * S) * es / 4
        return {'workload': note, 'precision': precision, 'recordings': n_rec, 'T': T, 'S': S, 'streams': nstreams,
                'ms_per_iteration': 1e3 * dt / K, 'value': n_rec * K / dt, 'unit': 'recording-EM-iterations/s',
                'blocks_of_K_steps': len(ts), 'seconds': sum(ts),
                'dominant_kernel': dk, 'avg_us': roof['avg_launch_us'], 'algorithmic_bytes': roof['algorithmic_bytes_per_launch'],
                'frac': roof['frac'], 'achieved_GBs': roof['achieved'], 'traffic': roof['traffic'],
                'traffic_source': roof['traffic_source'], 'bound': roof['bound'], 'bound_today': roof['bound_today'], 'simd_issue': roof['simd_issue'],
                'rho_copies_read': n_rec if rho_copies is None else rho_copies,
                'iteration_compulsory_bytes': fused, 'iteration_frac_of_hbm_peak': fused / (dt / K) / 1e9 / HBM_PEAK_GBS,
                'kernels_avg_us': {k: round(v['avg_us'], 2) for k, v in pk.items()}, 'elbo_last': elbo}

    configs = {}
    if rank == 0 and world == 1 and not args.no_configs:      # (a multi-GPU run measures the sharded batch only)
        def single(T, S, precision, lp):
            def make(mi, st):
                from vbx_amd.synth import make_recording
                b = _capi.Batch(ctx, [T], [S], args.D, precision=precision, max_iters=mi)
                X, Phi, _ = make_recording(T, S, D=args.D, seed=0, kappa=0.05, dtype=np.float32)
                g = np.random.default_rng(10_000).gamma(1.0, size=(T, S)).astype(np.float32)
                g /= g.sum(1, keepdims=True)
                b.set_recording(0, X, Phi, np.ones(S) / S, g, lp, 0.3, 17.0)
                return b
            return make
        short = args.min_seconds / 4
        # configs[0]: the reference's example recording, ONE reference-style call with host arrays in and out (allocation,
        # H2D of X, the loop with the reference's own stopping rule, the gamma write-out, D2H) -- call-level, PCIe inclusive
        gpath = os.path.join(REPO, 'tests', 'golden', 'es2005a.npz')
        if os.path.exists(gpath):
            import vbx_amd
            g = dict(np.load(gpath))
            kw = dict(pi=int(g['qinit'].shape[1]), gamma=g['qinit'], loopProb=float(g['loopProb']), Fa=float(g['Fa']), Fb=float(g['Fb']))
            n_ref = len(g['Li40'])
            for prec in ('fp64', 'fp32', 'fp32-split'):
                X = g['fea'] if prec == 'fp64' else g['fea'].astype(np.float32)
                call = dict(maxIters=40, epsilon=1e-6) if prec == 'fp64' else dict(maxIters=n_ref, epsilon=-1e300)
                ts = []
                for _ in range(12):
                    t0 = time.perf_counter()
                    q, sp, L = vbx_amd.VBx(X, g['Phi'], precision=prec, **kw, **call)
                    ts.append(time.perf_counter() - t0)
                dt = statistics.median(ts[2:])
                configs[f'C1_ES2005a_{prec}'] = {
                    'workload': 'configs[0]: exp/ES2005a x-vectors (T=1025, S=31 AHC clusters, R=128), VBx() as vbhmm.py:154-158 '
                                'calls it; fp64 under the reference\'s own stopping rule (maxIters=40, epsilon=1e-6), fp32 over the '
                                'same 13 iterations', 'precision': prec, 'iterations': len(L), 'reference_iterations': n_ref,
                    'ms_per_call': 1e3 * dt, 'value': len(L) / dt, 'unit': 'EM iterations/s (call-level: host arrays in and out)',
                    'gamma_max_abs_diff_vs_reference': float(np.abs(q - g['gamma40']).max()),
                    'elbo_rel_diff_vs_reference': float(abs(L[-1][0] - g['Li40'][-1]) / abs(g['Li40'][-1])),
                    'calls_timed': len(ts) - 2, 'note': 'median of 10 calls after 2 warm-up calls'}
        # configs[3] AS STATED: 64 recordings over 8 GPUs = 8 per GPU (the headline line holds 64 per GPU: weak scaling)
        for prec in ('fp32', 'fp32-split', 'fp64'):
            configs[f'C4_as_stated_8_per_gpu_{prec}'] = one_config(
                'C4', lambda mi, st, prec=prec: make_batch(ctx, 8, args.T, args.S, args.D, prec, 0, mi, st), 8, args.T, args.S, prec,
                'configs[3] as stated: 64 recordings over 8 GPUs = 8 recordings of T=10 000, S=30 on THIS GPU '
                '(python bench.py --gpus 8 --total-recordings 64 runs exactly that split)', short)
        # 64 < S <= 256 runs the chunked scan with operators in HBM, not the fused kernels: what an AHC
        # result of ~100 clusters on a long file costs (vbhmm.py:150-158)
        for prec in ('fp32', 'fp64'):
            configs[f'S128_T10k_{prec}'] = one_config('S128', single(10000, 128, prec, 0.99), 1, 10000, 128, prec,
                                                       'one recording, T=10 000, S=128: the wide chunked scan (64 < S <= 256), unfused kernels; kernel by kernel in '
                                                       'profiles/r04_s128_*_kernel_stats.txt: the largest piece is the boundary walk (fb_aux, scan2_wide: one '
                                                       'workgroup per direction, DESIGN section 11)', short)
        for prec in ('fp32', 'fp32-split', 'fp64'):
            configs[f'C2_T10k_S10_{prec}'] = one_config('C2', single(10000, 10, prec, 0.99), 1, 10000, 10, prec,
                                                         'configs[1]: one recording, T=10 000, S=10, Fa=0.3 Fb=17 loopProb=0.99', short)
            configs[f'C3_T50k_S30_{prec}'] = one_config('C3', single(50000, 30, prec, 0.99), 1, 50000, 30, prec,
                                                         'configs[2]: one recording, T=50 000, S=30 (two-level boundary walk)', short)
            for mode in ('shared', 'private'):
                if mode == 'private' and prec != 'fp32':
                    continue
                configs[f'C5_T200k_S50_sweep9_{prec}_{mode}'] = one_config(
                    f'C5_{mode}', lambda mi, st, prec=prec, mode=mode: make_sweep_batch(ctx, 200000, 50, args.D, prec, mi, mode == 'shared', streams=st),
                    len(SWEEP_POINTS), 200000, 50, prec,
                    'configs[4]: T=200 000, S=50, loopProb=0.9, the nine (Fa, Fb) points of the recipes as ONE batch, '
                    + ('one rho shared by all points (vbx_batch_set_recording_shared)' if mode == 'shared'
                       else 'every point with a private copy of the x-vectors (round 2)'), short,
                    rho_copies=1 if mode == 'shared' else None)

    other = None
    if other_prec is not None and not args.no_split:
        pko, dko, mso, _ = kernel_probe(lambda mi, st: headline(other_prec, mi, st), args.min_seconds / 4, args.max_blocks)
        bo = headline(other_prec, budget, args.streams)
        bo.run(W, -np.inf)
        to = timed_blocks(bo, args.min_seconds / 2, args.max_blocks)
        assert bo.gemm == ('split' if other_prec == 'fp32-split' else 'exact')
        bo.close()
        other = (to, pko, dko, mso)

    f64 = None
    if not args.no_f64 and args.precision == 'fp32':
        pk64, dk64, ms64, _ = kernel_probe(lambda mi, st: headline('fp64', mi, st), args.min_seconds / 4, args.max_blocks)
        b64 = headline('fp64', budget, args.streams)
        b64.run(W, -np.inf)
        t64 = timed_blocks(b64, args.min_seconds / 2, args.max_blocks)
        b64.close()
        f64 = (t64, pk64, dk64, ms64)

    single_rec = None
    if not args.no_single and rank == 0 and world == 1:
        b1 = make_batch(ctx, 1, args.T, args.S, args.D, args.precision, seed0=0, max_iters=W + 4 * K)
        b1.run(W, -np.inf)
        ts = []
        for _ in range(3):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            b1.run(K, -np.inf)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        dt = statistics.median(ts)
        single_rec = {'value': K / dt, 'unit': 'EM iterations/s', 'ms_per_iteration': 1e3 * dt / K, 'batch': 1}
        b1.close()

    call_rec = None
    if rank == 0 and world == 1 and not args.no_call_level:
        call_rec = call_level(args.batch, args.T, args.S, args.D, head_prec)

    if rank == 0:
        total_units = world * args.batch * K
        workload = {'batch': args.batch, 'T': args.T, 'S': args.S, 'D': args.D, 'precision': head_prec}
        roof, traffic = roofline_of(per_kernel, dom, args.batch, args.T, args.S, args.D, esize, workload)   # (profiled with --streams 1)
        roof['measured'] = (f'HIP events over {probe_blocks} block(s) of K steps (after W warm-up steps) of the same batch on ONE '
                            f'stream (VBX_OPT_STREAMS=1, {probe_ms_per_step:.4f} ms per step): a kernel-level figure needs the '
                            'kernel alone on the GPU')
        out = {
            'metric': 'VB EM iterations/sec (T=10k xvecs, R=128, S=30)',
            'value': total_units / med,
            'unit': 'recording-EM-iterations/s',
            'n_gpus': world,
            'steps': K,
            'warmup': W,
            'ms_per_step': 1e3 * med / K,
            'higher_is_better': True,
            'scaling': 'strong' if args.total_recordings is not None else 'weak',
            'vs_baseline': None,
            'dtype': ('f32 (f16-pair split products, f32 accumulate)' if args.gemm == 'split' else 'f32') if args.precision == 'fp32' else 'f64',
            'gemm': (args.gemm if args.precision == 'fp32' else 'exact'),
            'data': 'synthetic',
            'config': {'workload': f'batch of {args.batch} recordings per GPU, each T={args.T} x-vectors, '
                                   f'R={args.D}, S={args.S} (BASELINE configs[3] batch; headline shape), '
                                   'random gamma init, Fa=0.3 Fb=17 loopProb=0.99',
                       'recordings_per_gpu': args.batch, 'T': args.T, 'R': args.D, 'S': args.S, 'streams_per_gpu': streams,
                       'parallelism': f'recordings sharded over {world} rank(s), no data-path collective'
                                      + ('' if backend == 'nccl' or world == 1 else f' [TEST HOOK: backend {backend}, every rank on GPU {device} -- not a scaling measurement]')},
            'timed_region': {'blocks_of_K_steps': len(times), 'seconds': sum(times), 'statistic': 'median block',
                             'ms_per_step_min': 1e3 * min(times) / K, 'ms_per_step_max': 1e3 * max(times) / K,
                             'note': 'every block holds exactly K steps between barrier + synchronize; includes the '
                                     'one gamma write-out (replay launch) per block'},
            'device': info['name'],
            'kernels_avg_us': {k: round(v['avg_us'], 2) for k, v in per_kernel.items()},
            'kernels_avg_us_note': 'one-stream pass: the HBM-side kernels (ALGO_PASSES) over the K steps, the others over 8 survey iterations; '
                                   '"post" = the gamma write-out, once per run',
            'roofline': roof,
            'gamma_checks': {'row_sum_max_dev': float(np.abs(res0['gamma'].sum(1) - 1).max()),
                             'elbo_last': float(res0['Li'][-1])},
        }
        # what one recording keeps resident in HBM: rho, in split mode also its two f16-pair copies
        # (4 bytes per element each, made once per upload by rho_absmax + rho_split: two more passes over rho inside the first
        # run, under VBX_K_PREP); b and the caller's gamma [T][Sp]; per chunk the partial sums and the three operators
        Tp, Sp_ = -(-args.T // 128) * 128, 16 if args.S <= 16 else 32 if args.S <= 32 else 64
        chunks = Tp // 128
        out['hbm_resident_bytes_per_recording'] = {
            'rho': Tp * args.D * esize, 'rho_f16_pair_copies': (2 * Tp * args.D * 4) if head_prec == 'fp32-split' else 0,
            'b_and_gamma': 2 * Tp * Sp_ * esize, 'partials_and_operators': chunks * (Sp_ * args.D + 3 * Sp_ * Sp_ + 4 * Sp_) * esize}
        step_s = med / K
        e4 = esize / 4
        survey_bytes = args.batch * (8 * args.T * args.D + 28 * args.T * args.S) * e4
        fused_bytes = args.batch * (8 * args.T * args.D + 8 * args.T * args.S) * e4
        whole = {'survey_8d_bytes_per_step': survey_bytes,
                 'survey_8d_frac_of_hbm_peak': survey_bytes / step_s / 1e9 / HBM_PEAK_GBS,
                 'fused_compulsory_bytes_per_step': fused_bytes,
                 'fused_compulsory_GBs': fused_bytes / step_s / 1e9,
                 'fused_compulsory_frac_of_hbm_peak': fused_bytes / step_s / 1e9 / HBM_PEAK_GBS,
                 'note': 'survey_8d = 8TR + 28TS (one kernel per stage, SURVEY 8d); fused_compulsory = 8TR + 8TS: rho read '
                         'twice (the log-likelihoods need the alpha that needs the full-T reduction), b written and read '
                         'once, gamma and the lattices never leave the chip'}
        if traffic[2] and 'iteration_hbm_bytes' in traffic[2]:
            whole['pmc_bytes_per_step'] = traffic[2]['iteration_hbm_bytes']
            whole['pmc_over_fused_compulsory'] = traffic[2]['iteration_hbm_bytes'] / fused_bytes
        out['roofline_whole_iteration'] = whole
        if other:
            to, pko, dko, mso = other
            mo = statistics.median(to)
            roofo, _ = roofline_of(pko, dko, args.batch, args.T, args.S, args.D, 4, dict(workload, precision=other_prec))
            key = 'f32_split' if other_prec == 'fp32-split' else 'f32_exact'
            out[key] = {'value': total_units / mo, 'unit': 'recording-EM-iterations/s', 'ms_per_step': 1e3 * mo / K,
                        'blocks_of_K_steps': len(to), 'roofline': roofo, 'one_stream_ms_per_step': mso,
                        'kernels_avg_us': {k: round(v['avg_us'], 2) for k, v in pko.items()},
                        'speedup_over_headline': med / mo,
                        'note': ('same batch, fp32 storage and accumulation, the two GEMMs (rho alpha^T, gamma^T rho) on '
                                 'v_mfma_f32_16x16x32_f16 with error-compensated f16 operand pairs (VBX_OPT_GEMM = split, '
                                 'vbx_amd/csrc/vbx_split.hpp); parity against the reference under the bounds of the exact path: '
                                 'tests/test_gpu_split.py, tests/test_gpu_configs.py [fp32-split], profiles/*config_parity.json')
                                if other_prec == 'fp32-split' else
                                'same batch with the exact f32 GEMMs (v_mfma_f32_16x16x4_f32)'}
        if f64:
            t64, pk64, dk64, ms64 = f64
            m64 = statistics.median(t64)
            w64 = dict(workload, precision='fp64')
            roof64, _ = roofline_of(pk64, dk64, args.batch, args.T, args.S, args.D, 8, w64)
            out['f64'] = {'value': total_units / m64, 'unit': 'recording-EM-iterations/s', 'ms_per_step': 1e3 * m64 / K,
                          'blocks_of_K_steps': len(t64), 'roofline': roof64,
                          'kernels_avg_us': {k: round(v['avg_us'], 2) for k, v in pk64.items()},
                          'one_stream_ms_per_step': ms64,
                          'note': 'same batch on the fp64 path (f64 storage, v_mfma_f64_16x16x4_f64): what vbhmm.py '
                                  'gets, its inputs being float64; reproduces the reference\'s iteration counts'}
        if strong:
            ms_ = statistics.median(strong[0])
            out['strong_scaling_form'] = {
                'value': STATED_TOTAL * K / ms_, 'unit': 'recording-EM-iterations/s', 'ms_per_step': 1e3 * ms_ / K,
                'scaling': 'strong', 'recordings_total': STATED_TOTAL, 'recordings_per_gpu': STATED_TOTAL // world,
                'streams_per_gpu': strong[1], 'blocks_of_K_steps': len(strong[0]),
                'note': 'BASELINE configs[3] as stated: 64 recordings over all ranks (python bench.py --gpus N '
                        '--total-recordings 64 makes this the headline); 8 per GPU at N = 8 is the launch-latency regime'}
        if configs:
            out['configs'] = configs
            out['configs_note'] = ('the other configurations of BASELINE.json, each measured like the headline: ms per EM iteration '
                                   'over blocks of K steps (median), value = recording-iterations/s, and the dominant HBM-side '
                                   'kernel of a one-stream pass with its algorithmic bytes / HIP-event time vs the 8 TB/s peak; '
                                   'C5 counts nine recording-iterations per sweep iteration')
        if single_rec:
            out['single_recording'] = single_rec
        if call_rec:
            out['call_level'] = call_rec
        if world == 1 and args.cpu_iters > 0:
            cb = cpu_baseline(args.T, args.S, args.D, args.cpu_iters, head_prec, args.parity_iters, args.cpu_reference)
            out['cpu_baseline'] = cb
            if single_rec:
                out['single_recording']['speedup_vs_cpu_baseline'] = single_rec['value'] / cb['value']
        emit(out, args.full_out)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
