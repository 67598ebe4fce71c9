"""Host-side orchestration of the gfx950 Tacotron 2 hot path (teacher-forced forward / backward, optimizer).

Everything numerical is a call into libtacotron2_amd.so (include/tacotron2_amd.h); torch is used only to own
device memory and the stream.  The forward follows model/tacotron2.py:155-347 of the reference, re-scheduled for
the hardware:

  * every nn.Linear / nn.Conv1d with a full (batch x time) row block is ONE large fp32-MFMA GEMM
    (conv = GEMM over overlapping channel-last rows), including the parts of the two LSTMCells' input
    projections that do not depend on the recurrence (prenet -> att_rnn, [att_h, ctx] -> decoder lstm);
  * in teacher-forced mode the attention recurrence (att_rnn + attention) does not depend on the decoder
    LSTM, so the frame loop runs as two chains: attention chain (3 launches / frame) then decoder-LSTM chain
    (1 launch / frame), and the mel/stop projection becomes one GEMM over all frames.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

from . import _lib
from ._lib import call, make
from .params import ParamStore

KL, KPAD = 31, 15


def _stream():
    return torch.cuda.current_stream().cuda_stream


SHARE_CU = [0]     # set to [1] while enqueuing GEMMs that run next to a latency-bound chain on another stream
# 0: fp32 GEMMs on the bf16 matrix pipe by error-free 3-way operand splitting (csrc/t2_gemm.hip, fp32-accurate, 2.7x the MFMA
# rate); 1: f32-input MFMA.  One process-wide switch for A/B measurements and the kernel tests.
import os as _os
GEMM_NATIVE_FP32 = [int(_os.environ.get("T2_GEMM_NATIVE_FP32", "0"))]
# torch.set_float32_matmul_precision of the reference (run/train.py:170, every shipped config says "high"): how many of the bf16
# partial products the GEMMs issue.  "highest" (default here; what bench.py and the parity tests measure) = fp32-exact operands.
MATMUL_PRECISION = {"highest": 0, "high": 1, "medium": 2}
GEMM_PRECISION = [0]


def set_float32_matmul_precision(name: str) -> None:
    if name not in MATMUL_PRECISION:
        raise ValueError(f"float32_matmul_precision must be one of {sorted(MATMUL_PRECISION)}, got {name!r}")
    GEMM_PRECISION[0] = MATMUL_PRECISION[name]


def gemm(A, B, C, M, N, K, lda, ldb, ldc, a_k=1, b_k=1, alpha=1.0, bias=None, bias2=None, mulmask=None, ldmask=0,
         relu=0, accumulate=0, splitk=1, batch=1, sA=0, sB=0, sC=0, a_tap_len=0, a_tap_stride=0, stat_out=None, stat_Lp=0, stat_L=0):
    """C-ABI t2_gemm on raw pointers (ints) or tensors."""
    g = make("T2Gemm", A=A, B=B, C=C, M=M, N=N, K=K, lda=lda, ldb=ldb, ldc=ldc, a_kmajor=a_k, b_kmajor=b_k,
             alpha=alpha, bias=bias, bias2=bias2, mulmask=mulmask, ldmask=ldmask, relu=relu, accumulate=accumulate,
             splitk=splitk, batch=batch, sA=sA, sB=sB, sC=sC, share_cu=SHARE_CU[0], native_fp32=GEMM_NATIVE_FP32[0], precision=GEMM_PRECISION[0],
             a_tap_len=a_tap_len, a_tap_stride=a_tap_stride, stat_out=stat_out, stat_Lp=stat_Lp, stat_L=stat_L)
    call("t2_gemm", g, _stream())


_PREZEROED: Dict[int, bool] = {}      # data_ptr of buffers a phase's prologue has already put on the zero list (one-shot marks)


def fill_splits(M: int, N: int, K: int) -> bool:
    return ((M + 127) // 128) * ((N + 127) // 128) < 256 and K >= 1024


def gemm_fill(A, B, C, M, N, K, lda, ldb, ldc, a_k=1, b_k=1, bias=None):
    """Plain C = A.B (+ bias) for shapes that leave the chip under-filled (fewer than 256 output tiles of 128 x 128, long K): two
    K slices accumulate into the zeroed C with atomics - encoder convolution 178 -> 123 us, BiLSTM dgrad 123 -> 94 us
    (tools/bench_gemm_small.py).  C must be a tensor whose first M rows of ldc floats are exactly the output."""
    if fill_splits(M, N, K) and torch.is_tensor(C):
        if not _PREZEROED.pop(C.data_ptr(), False):      # (cleared ahead, with the phase's other regions: Engine.prezero)
            zero_later(C.view(-1)[:M * ldc])
        gemm(A, B, C, M, N, K, lda, ldb, ldc, a_k=a_k, b_k=b_k, bias=bias, accumulate=2, splitk=2)
    else:
        gemm(A, B, C, M, N, K, lda, ldb, ldc, a_k=a_k, b_k=b_k, bias=bias)


def _ptr(t: torch.Tensor, elem_off: int = 0) -> int:
    return t.data_ptr() + 4 * elem_off


# ---- deferred zeroing: ONE launch for the state tensors a phase clears ----------------------------------------------------------
# The reference clears its recurrent state with one ATen fill per tensor (model/tacotron2.py:126-153); the engine has ~55 such
# regions per training step (slot 0 of every time-major stash, split-K accumulators, gradient accumulators, BatchNorm sums, the
# arrival counters of the persistent launches, the flat gradient buffer).  zero_later() only RECORDS a region, keyed by the torch
# stream that was current; the list of every stream is cleared by one t2_zero_regions launch on that stream in front of the next
# library call of the process (a hook in _lib.call), i.e. before anything enqueued later can read or accumulate into it.  Code that
# reads such a buffer with a torch operation and no library call in between must call flush_zeros() itself.
# Per THREAD: a loader thread that calls into the library (log-mel) must not flush - or delay - the training thread's list.
import threading as _threading
_TLS = _threading.local()


def _pending() -> Dict[int, list]:
    d = getattr(_TLS, "pending", None)
    if d is None:
        d = _TLS.pending = {}
    return d


def zero_later(t: torch.Tensor) -> torch.Tensor:
    """Record `t` (contiguous, or 2-D with unit inner stride) for the next t2_zero_regions launch on the current stream."""
    if t.numel() == 0:
        return t
    es = t.element_size()
    if t.is_contiguous():
        reg = (t.data_ptr(), t.numel() * es, 1, t.numel() * es)
    else:
        assert t.dim() == 2 and t.stride(1) == 1, "zero_later: contiguous tensors or rows with unit inner stride"
        reg = (t.data_ptr(), t.shape[1] * es, t.shape[0], t.stride(0) * es)
    assert reg[0] % 4 == 0 and reg[1] % 4 == 0 and reg[3] % 4 == 0, "zero_later: regions are runs of whole 4-byte words"
    st = torch.cuda.current_stream(t.device)
    _pending().setdefault(st.cuda_stream, []).append((reg, t))       # (the tensor is kept alive until the launch is enqueued)
    return t


def flush_zeros() -> None:
    mine = _pending()
    if not mine:
        return
    pending = list(mine.items())
    mine.clear()                               # (first: the t2_zero_regions calls below come back through the hook)
    for stream, regs in pending:
        for i0 in range(0, len(regs), 64):
            part = regs[i0:i0 + 64]
            z = _lib.S["T2ZeroRegions"]()
            for i, ((ptr, rb, nr, sb), _) in enumerate(part):
                z.p[i] = ptr; z.row_bytes[i] = rb; z.nrows[i] = nr; z.stride_bytes[i] = sb
            z.n = len(part)
            call("t2_zero_regions", z, stream)


_lib.PRE_CALL.append(flush_zeros)


def splitk_for(M: int, N: int, K: int) -> int:
    """K-slices for weight-gradient shaped GEMMs (few output tiles, very long K).  Cost model fitted to
    profiles/r01_gemm_shapes.txt: 128x128 tiles, 256 CUs, the kernel is MFMA-bound per CU, so the time is
    ceil(workgroups / 256) rounds of 1/sk tile-times; fewer than ~1.5 workgroups per CU hides latency worse (+15 %);
    every extra slice adds atomic traffic (+1 %)."""
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    kt = (K + 31) // 32
    best, best_cost = 1, None
    for sk in range(1, max(1, min(kt // 4, 64)) + 1):
        w = tiles * sk
        cost = ((w + 255) // 256) / sk * (1.15 if w < 384 else 1.0) * (1.0 + 0.01 * sk)
        if best_cost is None or cost < best_cost - 1e-12:
            best, best_cost = sk, cost
    return best


def _chunk_sizes(T: int, CH: int, ramp_at_end: bool = True):
    """Frame counts of the pipeline chunks in time order: CH each, and - with the ramp - the last CH frames as CH/2, CH/4, CH/8,
    CH/8 (so the pipeline's fill / drain, which is not overlapped, is an eighth of a chunk)."""
    if not ramp_at_end or CH < 16 or T <= CH:
        return [min(CH, T - c0) for c0 in range(0, T, CH)]
    tail = [CH // 2, CH // 4, CH // 8, CH - CH // 2 - CH // 4 - CH // 8]
    body = T - CH
    sizes = [min(CH, body - c0) for c0 in range(0, body, CH)]
    return sizes + tail


class Engine:
    """Forward/backward of the whole model on one device.  `ps` is the ParamStore."""

    def __init__(self, ps: ParamStore):
        self.ps = ps
        self.d = ps.dims
        self.dev = ps.device
        self._ws: Dict[str, torch.Tensor] = {}
        self._prez: Dict[str, bool] = {}
        self._bn_clean: set = set()   # BatchNorm workspace slots cleared by begin_phase and not used since
        self._persist_next = 0
        self._side = None
        self.chunk = 64               # frames per pipeline chunk of the forward frame loop
        self.chunk_bwd = 64           # frames per chunk of the backward pipeline (r03: 64 beats 80 by 0.2 ms with the BPTT launches at default wave priority)
        self.dec_chain = "persistent" # forward decoder-LSTM chain: "persistent" (one weight-stationary launch per chunk on the side
                                      # stream) or "steps" (one launch per frame there; also what runs when the persistent launch
                                      # cannot be co-resident or the batch has more than 64 rows)
        self.persist_gemm_side = True    # the hoisted pre_dec GEMM of a chunk runs on the side stream too, in front of the chunk's
                                         # persistent launch (70.0 against 71.3 ms per step on the main stream, profiles/r02_ab_fwd_dec_chain.txt)
        self._persist_sync = None
        self._persist_ok = {}            # (D, one row tile) -> residency verdict of the persistent launch
        self.defer_wgrads = True      # weight-gradient GEMMs of postnet, projection and encoder leave the main stream (-1.4 ms per step)
        self.wgrad_group = 4          # pipeline chunks per weight-gradient GEMM call (profiles/r02_ab_wgrad_pipeline.txt)
        self.chunk_att_wgrads = True  # attention-chain weight gradients per pipeline chunk, behind the chain, instead of all at its end
        self.ramp_chunks = True       # short chunks at the un-overlapped end of the forward / start of the backward pipeline
        self.share_cu = 1             # side-stream GEMMs next to the chains at ONE workgroup per CU: two 73 KB-LDS workgroups
                                      # per CU lock the attention kernels out (84.3 -> 83.2 ms)
        self.splitk_small_chunks = True  # forward pipeline: the hoisted decoder-LSTM input GEMM of short chunks runs split-K
        self.enc_chain = "persistent"    # encoder BiLSTM recurrence: "persistent" (one launch, both directions) | "steps" (S launches)
        self.enc_persist_max_rows = 32   # (the launch itself takes up to 64 rows, as two consecutive blocks)
        self.bn_epilogue_stats = False  # BatchNorm statistics from the producing GEMM's epilogue (T2Gemm.stat_out) instead of their own pass:
                                        # built and measured in round 5 - more accurate (per-tile shifts), but the GEMM is VALU-bound and
                                        # its epilogue costs more than the 22 us statistics pass it saves (profiles/r05_ab_bn_epilogue_stats.txt)
        self.bptt_off_chain = True    # the decoder-LSTM BPTT launches (side stream, a chunk ahead) keep the default wave priority
        self.sync_bn_group = None     # torch.distributed group: BatchNorm statistics over all ranks' shards (Trainer(sync_bn=True))
        self.grad_tail_hook = None    # called on the side stream once the gradients from prenet.0.weight onwards are enqueued
        self.generation = 0           # bumped by every forward: activations live in the shared named workspaces,
                                      # so only the LATEST forward can be back-propagated (checked in backward_tf)
        self.profile = False          # when True, mark() records HIP events at segment boundaries
        self.marks = []; self.spans = []               # [(name, event)] of the current step

    def persist_resident(self, D: int, B: int, n: int = 1) -> bool:
        """True when the persistent LSTM launch (n cells x H/4 workgroups that wait for each other) is fully co-resident on this
        device (t2_lstm_persist_resident: compute units x occupancy); otherwise the chain runs as per-step launches."""
        key = (D, min(B, 32) <= 16, n)
        if key not in self._persist_ok:
            rc = _lib.call_value("t2_lstm_persist_resident_n", D, D, min(B, 32), n)
            if rc not in (0, 3):
                raise _lib.T2Error(f"t2_lstm_persist_resident failed (rc={rc}): {_lib.lib().t2_last_error().decode()}")
            self._persist_ok[key] = rc == 0
        return self._persist_ok[key]

    def persist_sync(self):
        """Scratch of the persistent launches: arrival counters (words 0..255), sticky timeout flag (word 256)."""
        if self._persist_sync is None:
            self._persist_sync = torch.zeros(self.PERSIST_RING0 + 256 * self.PERSIST_RING, dtype=torch.int32, device=self.dev)
        return self._persist_sync

    def check_persistent_kernels(self):
        """Host-synchronising check of the persistent launches' timeout flag (bounded spins end the launch early instead of
        hanging): raises if a wait timed out.  Tests and the bench call it after a step."""
        if self._persist_sync is not None and int(self._persist_sync[256].item()) != 0:
            self._persist_sync[256:272].zero_()       # the flag is sticky on the device: cleared here, reported once
            raise _lib.T2Error("t2_lstm_seq_fwd_persist: an inter-workgroup wait timed out (outputs of that forward are invalid)")

    def side_stream(self):
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.dev)
        return self._side

    def stream_concurrency_check(self, n: int = 200, spin_us: int = 3000) -> dict:
        """Do this engine's two streams run side by side?  One idle wave holds the SIDE stream for `spin_us` (t2_stream_probe_spin)
        while `n` dependent one-thread launches run on the MAIN stream (t2_stream_probe_chain); HIP events on the main stream around
        the whole pattern, the first one recorded BEFORE the spin starts (the side stream waits for it).  Streams that the runtime
        mapped onto ONE hardware queue serialise - the chain then ends `spin_us` later than it does alone.  (Measured, round 5,
        profiles/r05_queue_check_probe.txt: GPU_MAX_HW_QUEUES=1, or 4 with a live RCCL communicator in the process, puts both
        streams behind one queue and the training step takes 87 instead of 64 ms.  Events recorded AFTER the spin was enqueued
        do not see it: in a shared queue they wait behind the spin too, and the chain between them even looks faster.)
        Synchronises the host; meant for start-up (Trainer calls ensure_concurrent_streams once when data-parallel)."""
        import os
        main, side = torch.cuda.current_stream(), self.side_stream()
        w = torch.zeros(8, dtype=torch.int32, device=self.dev)

        def pattern(spin):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(main)
            if spin:
                side.wait_event(e0)
                call("t2_stream_probe_spin", _ptr(w, 4), spin, side.cuda_stream)
            call("t2_stream_probe_chain", w, n, main.cuda_stream)
            e1.record(main)
            torch.cuda.synchronize(self.dev)
            return e0.elapsed_time(e1) * 1e3

        pattern(10)                                                  # warm-up (code objects, first use of the side stream)
        alone_us = pattern(0)
        beside_us = pattern(spin_us)
        ok = beside_us < alone_us + 0.5 * spin_us
        return dict(ok=bool(ok), launches=n, chain_alone_us=round(alone_us, 1), chain_beside_spin_us=round(beside_us, 1),
                    spin_us=spin_us, us_per_dependent_launch=round(alone_us / n, 2),
                    GPU_MAX_HW_QUEUES=os.environ.get("GPU_MAX_HW_QUEUES"))

    def ensure_concurrent_streams(self, max_tries: int = 8) -> dict:
        """stream_concurrency_check, and a remedy when it fails: a stream is bound to a hardware queue when it is created, so a
        NEW side stream may land on another queue than the main stream's (the old ones are kept alive - the runtime hands the
        least-used queue to the next stream).  Returns the last check with `tries`; `ok` False after max_tries means every queue
        the runtime offers is shared with the main stream (GPU_MAX_HW_QUEUES=1)."""
        qc = self.stream_concurrency_check()
        tries = 1
        while not qc["ok"] and tries < max_tries:
            self._spare_streams = getattr(self, "_spare_streams", []) + [self._side]
            self._side = torch.cuda.Stream(device=self.dev)
            qc = self.stream_concurrency_check()
            tries += 1
        qc["tries"] = tries
        return qc

    def mark(self, name: str):
        if self.profile:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream())
            self.marks.append((name, ev))

    def segment_times_ms(self):
        """Durations between consecutive marks of the last profiled step: {segment: ms} (summed per name), plus the
        side-stream spans recorded with mark_span (work that runs concurrently with main-stream segments)."""
        out: Dict[str, float] = {}
        for (n0, e0), (n1, e1) in zip(self.marks[:-1], self.marks[1:]):
            out[n1] = out.get(n1, 0.0) + e0.elapsed_time(e1)
        for name, e0, e1 in self.spans:
            out[name] = out.get(name, 0.0) + e0.elapsed_time(e1)
        return out

    def span_begin(self):
        if not self.profile:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(torch.cuda.current_stream())
        return ev

    def span_end(self, name: str, e0):
        if e0 is not None:
            ev = torch.cuda.Event(enable_timing=True)
            ev.record(torch.cuda.current_stream())
            self.spans.append((name, e0, ev))

    def make_masks(self, B: int, L: int, T: int, training: bool, seed: int, step: int) -> dict:
        """Dropout scale masks for one step from the device Philox generator (t2_philox_mask).  Sites and rates as
        the reference: encoder/postnet p, prenet p always on (model/modules.py), LSTMCell outputs 0.1 (model/decoder.py:29,43)."""
        d = self.d
        p = float(d["dropout"])
        E, Pd, A, D, M, Pn = d["encoded_dim"], d["prenet_dim"], d["att_rnn_dim"], d["rnn_hidden_dim"], d["num_mels"], d["postnet_dim"]
        st = _stream()
        sid = [step * 64]

        def gen(name, n, rate):
            m = self.buf("mask." + name, n)
            call("t2_philox_mask", m, n, rate, seed, sid[0], st)
            sid[0] += 1
            return m

        masks = {}
        if p > 0.0:
            masks["prenet_drop"] = [gen(f"pre{i}", (T + 1) * B * Pd, p).view(T + 1, B, Pd) for i in range(2)]
        if training:
            if p > 0.0:
                masks["enc_drop"] = [gen(f"enc{i}", B * L * E, p).view(B, L, E) for i in range(3)]
                chans = [Pn, Pn, Pn, Pn, M]
                masks["post_drop"] = [gen(f"post{i}", B * T * c, p).view(B, T, c) for i, c in enumerate(chans)]
            masks["att_drop"] = gen("att", T * B * A, 0.1).view(T, B, A)
            masks["dec_drop"] = gen("dec", T * B * D, 0.1).view(T, B, D)
        return masks

    # ---- workspace --------------------------------------------------------------------------------
    def buf(self, name: str, *shape, dtype=torch.float32, zero: bool = False) -> torch.Tensor:
        n = 1
        for s in shape:
            n *= s
        t = self._ws.get(name)
        if t is None or t.numel() < n or t.dtype != dtype:
            t = torch.empty(max(n, 1), dtype=dtype, device=self.dev)
            self._ws[name] = t
        v = t[:n].view(*shape)
        if zero and not self._prez.pop(name, False):
            zero_later(v)       # cleared by the next t2_zero_regions launch, in front of the next library call
        return v

    def workspace_report(self, top: int = 8) -> dict:
        """Device memory of the named workspaces as allocated so far (after a step at the largest shape seen): total bytes, count
        and the largest ones - the engine's part of the HBM footprint (the tanh stash of the attention backward dominates)."""
        sizes = sorted(((t.numel() * t.element_size(), n) for n, t in self._ws.items()), reverse=True)
        return dict(total_bytes=sum(b for b, _ in sizes), buffers=len(sizes), largest={n: b for b, n in sizes[:top]})

    def prezero(self, name: str, *shape, dtype=torch.float32) -> torch.Tensor:
        """Allocate the named workspace and put it on the zero list NOW (a phase's prologue: one launch clears everything the phase
        accumulates into); the later `buf(name, ..., zero=True)` of the code that uses it is then a plain lookup."""
        v = self.buf(name, *shape, dtype=dtype)
        zero_later(v)
        self._prez[name] = True
        return v

    BN_SLOTS = {"enc.conv0": 0, "enc.conv1": 1, "enc.conv2": 2, "post.conv0": 3, "post.conv1": 4, "post.conv2": 5, "post.conv3": 6,
                "post.conv4": 7}

    def bn_sums(self, tag: str, backward: bool) -> torch.Tensor:
        """The statistics workspace (2C + 2 doubles) of one BatchNorm layer: a slot of ONE arena that begin_phase clears for the
        phase's 8 layers at once (T2Bn.sums_prezeroed) instead of one memset per layer.  A slot is good for ONE use per clear: a
        layer driven without begin_phase (encoder_fwd / conv_bn_fwd called on their own), or twice in a phase, gets its slot put on
        the zero list right here."""
        C = max(self.d["encoded_dim"], self.d["postnet_dim"], self.d["num_mels"])
        arena = self.buf("bn.sums", 16, 2 * C + 2, dtype=torch.float64)
        if tag not in self.BN_SLOTS:      # a layer driven from outside the engine's phases (model/submodules.py): its own workspace
            return zero_later(self.buf(f"{tag}.sums", 2 * C + 2, dtype=torch.float64))
        slot = self.BN_SLOTS[tag] + (8 if backward else 0)
        if slot in self._bn_clean:
            self._bn_clean.discard(slot)
        else:
            zero_later(arena[slot])
        return arena[slot]

    def begin_phase(self, backward: bool):
        """Start of a forward (teacher-forced or inference) / of a backward: the BatchNorm sums of the phase's 8 layers and - forward
        - the arrival-counter ring of the persistent launches go on the zero list."""
        C = max(self.d["encoded_dim"], self.d["postnet_dim"], self.d["num_mels"])
        arena = self.buf("bn.sums", 16, 2 * C + 2, dtype=torch.float64)
        zero_later(arena[8:] if backward else arena[:8])
        self._bn_clean = (self._bn_clean - set(range(0, 16))) | set(range(8, 16) if backward else range(0, 8))
        if not backward:
            if self._persist_sync is not None:      # (created - zero-filled - by the first persistent launch of this engine)
                zero_later(self._persist_sync[self.PERSIST_RING0:])
            self._persist_next = 0

    PERSIST_RING0 = 320              # words 0..255 legacy counters, 256 sticky flag; from 320: ring of 256-word counter blocks
    PERSIST_RING = 96

    def persist_flag(self) -> int:
        return self.persist_sync().data_ptr() + 4 * 256

    def persist_counters(self, nblocks: int) -> int:
        """Device address of `nblocks` fresh 256-word arrival-counter blocks (cleared by begin_phase; one per row block of a
        persistent launch: t2_lstm_seq_fwd_persist_pz)."""
        assert self._persist_next + nblocks <= self.PERSIST_RING, "persistent launches per forward exceed the counter ring"
        a = self.persist_sync().data_ptr() + 4 * (self.PERSIST_RING0 + 256 * self._persist_next)
        self._persist_next += nblocks
        return a

    # ---- lane-contiguous weight streams for the LSTM step kernels (re-laid once per step) ------------
    def pack_fwd(self, name, segs, H):
        """segs: [(weight pointer, ld, K)], returns the packed buffer (t2_lstm_pack_fwd)."""
        arr = (_lib.S["T2Seg"] * len(segs))()
        ktot = 0
        for i, (w, ld, K) in enumerate(segs):
            arr[i].w = w if isinstance(w, int) else w.data_ptr()
            arr[i].ldw = ld; arr[i].K = K
            ktot += K
        ntpad = (ktot // 16 + 15) // 16 * 16
        out = self.buf("pack." + name, H // 4 * ntpad * 256)
        call("t2_lstm_pack_fwd", arr, len(segs), H, out, _stream())
        return out

    def pack_bwd(self, name, W, ldw, N4, ncols, W2=None, ldw2=0, N2=0):
        tiles = (ncols + 15) // 16
        nchpad = ((N4 + N2) // 16 + 31) // 32 * 32
        out = self.buf("pack." + name, tiles * nchpad * 256)
        call("t2_lstm_pack_bwd", W, ldw, N4, W2, ldw2, N2, ncols, out, _stream())
        return out

    # ---- encoder ----------------------------------------------------------------------------------
    def conv_bn_fwd(self, tag, x_pad, w, bias, bn_prefix, B, L, Ci, Co, act, drop, training, ctx,
                    y=None, Lp_y=None, pad_y=2, res=None, Lp_res=0, pad_res=0, length=None, fill=0.0):
        """x_pad (B, L+4, Ci) padded layout -> y (padded layout by default).  Saves raw conv output + stats."""
        P, Bf = self.ps.P, self.ps.Bf
        Lp = L + 4
        wp = self.buf(f"{tag}.wp", Co, 5 * Ci)
        call("t2_pack_conv_weight", w, wp, Co, Ci, 5, 0, _stream())
        raw = self.buf(f"{tag}.raw", B * Lp, Co)
        Mg = B * Lp - 4
        # BatchNorm batch statistics as per-tile partial sums in the convolution GEMM's epilogue (no extra pass over `raw`, no
        # atomics; merged by t2_bn_fwd) - wherever the GEMM is ONE plain pass over K (the under-filled encoder convolutions run
        # split-K with atomics instead, their statistics stay a kernel of their own)
        tstats = None
        if training and self.bn_epilogue_stats and not fill_splits(Mg, Co, 5 * Ci) and not GEMM_NATIVE_FP32[0]:
            tstats = self.buf(f"{tag}.tstats", (Mg + 127) // 128, 3, Co)
            gemm(x_pad, wp, raw, Mg, Co, 5 * Ci, Ci, 5 * Ci, Co, bias=bias, stat_out=tstats, stat_Lp=Lp, stat_L=L)
        else:
            gemm_fill(x_pad, wp, raw, Mg, Co, 5 * Ci, Ci, 5 * Ci, Co, bias=bias)
        mean = self.buf(f"{tag}.mean", Co)
        invstd = self.buf(f"{tag}.invstd", Co)
        sums = self.bn_sums(tag, backward=False)
        if y is None:
            y = self.buf(f"{tag}.y", B, Lp, Co)
            Lp_y = Lp
        sync = training and self.sync_bn_group is not None
        bn = make("T2Bn", B=B, L=L, C=Co, x=raw, Lp_x=Lp, gamma=P[bn_prefix + ".weight"], beta=P[bn_prefix + ".bias"],
                  running_mean=Bf[bn_prefix + ".running_mean"], running_var=Bf[bn_prefix + ".running_var"],
                  training=1 if training else 0, momentum=0.1, eps=1e-5, sums=sums, mean=mean, invstd=invstd, act=act,
                  drop=drop, res=res, Lp_res=Lp_res, pad_res=pad_res, len=length, fill=fill, y=y, Lp_y=Lp_y, pad_y=pad_y,
                  sums_prezeroed=1, tile_stats=tstats, tile_M=Mg if tstats is not None else 0)
        if sync:
            # synchronised batch statistics: every rank sums (x - shift), (x - shift)^2 and its row count, ONE all-reduce of
            # 2C + 2 doubles per layer, then every rank normalises with the statistics of the global batch (what the
            # single-device reference computes over 8 x 32 = 256 utterances).  The shift must be the same on every rank:
            # a copy of the running mean taken BEFORE this step's update.
            shift = self.buf(f"{tag}.shift", Co)
            shift.copy_(Bf[bn_prefix + ".running_mean"])
            bn.shift = shift.data_ptr(); bn.phase = 1
            call("t2_bn_fwd", bn, _stream())
            import torch.distributed as dist
            dist.all_reduce(sums, group=self.sync_bn_group)
            bn.phase = 2
        call("t2_bn_fwd", bn, _stream())
        if training:
            self.ps.num_batches_tracked[bn_prefix + ".num_batches_tracked"] += 1
        ctx[tag] = dict(x_pad=x_pad, raw=raw, mean=mean, invstd=invstd, drop=drop, wp=wp)
        return y

    def encoder_fwd(self, chars_idx, chars_len32, training, masks, ctx):
        d, P = self.d, self.ps.P
        B, L = chars_idx.shape
        E, H = d["encoded_dim"], d["encoded_dim"] // 2
        Lp = L + 4
        x = self.buf("enc.x0", B, Lp, E)
        if fill_splits(B * Lp - 4, E, 5 * E):      # (gemm_fill: the convolutions accumulate two K slices into a cleared output)
            for li in range(3):
                raw = self.buf(f"enc.conv{li}.raw", B * Lp, E)
                zero_later(raw.view(-1)[:(B * Lp - 4) * E]); _PREZEROED[raw.data_ptr()] = True
        call("t2_embedding_fwd", chars_idx, P["encoder.embedding.weight"], x, B, L, E, 2, _stream())
        enc_drop = masks.get("enc_drop") if masks else None
        for li, i in enumerate((0, 4, 8)):
            x = self.conv_bn_fwd(f"enc.conv{li}", x, P[f"encoder.convolutions.{i}.weight"],
                                 P[f"encoder.convolutions.{i}.bias"], f"encoder.convolutions.{i + 1}", B, L, E, E, 1,
                                 enc_drop[li] if enc_drop is not None else None, training, ctx)
        # BiLSTM: one input GEMM for both directions (N = 8H), then L steps with both directions per launch
        pre = self.buf("enc.pre", B * Lp, 8 * H)
        w_ih = self.ps.cat_view("encoder.lstm.weight_ih_l0", 8 * H, E)
        b_ih = self.ps.cat_view("encoder.lstm.bias_ih_l0", 8 * H, 0)
        b_hh = self.ps.cat_view("encoder.lstm.bias_hh_l0", 8 * H, 0)
        gemm(_ptr(x, 2 * E), w_ih, pre, B * Lp - 4, 8 * H, E, E, E, 8 * H, bias=b_ih, bias2=b_hh)
        S = L
        hs = self.buf("enc.h", 2, S + 1, B, H)
        cs = self.buf("enc.c", 2, S + 1, B, H)
        gs = self.buf("enc.gates", 2, S, B, 4 * H)
        for z in (hs[0, 0], cs[0, 0], hs[1, S], cs[1, S]):
            zero_later(z)
        enc = self.buf("enc.out", B, L, E)
        steps = (_lib.S["T2LstmStep"] * 2)()
        incs = (_lib.S["T2LstmStride"] * 2)()
        # The recurrence as ONE persistent launch for both directions (t2_lstm_seq_fwd_persist_n: 2 x H/4 workgroups, W_hh slices in
        # LDS, h exchanged through an x16-tiled stash) instead of S launches of ~8 us
        # (up to 32 rows: above, the persistent launch runs once per block of 32 rows and the step kernel's four row tiles win)
        persist = B <= self.enc_persist_max_rows and self.enc_chain == "persistent" and H % 16 == 0 and self.persist_resident(H, B, 2)
        ctx["enc_persist"] = persist
        if persist:
            Bp = (B + 15) // 16 * 16
            ht = self.buf("enc.ht", 2, S + 1, H // 16, Bp, 16, zero=(B != Bp))
            zero_later(ht[0, 0]); zero_later(ht[1, S])
        for dr in range(2):
            t0 = 0 if dr == 0 else S - 1
            sg = 1 if dr == 0 else -1
            whh = P["encoder.lstm.weight_hh_l0" + ("" if dr == 0 else "_reverse")]
            slot_in = 0 if dr == 0 else S          # slot holding the previous state
            slot_out = 1 if dr == 0 else S - 1
            st = steps[dr]
            st.B, st.H, st.nseg = B, H, 1
            st.wpacked = self.pack_fwd(f"enc.whh{dr}", [(whh, H, H)], H).data_ptr()
            st.seg[0].x = _ptr(hs[dr, slot_in]); st.seg[0].ldx = H
            st.seg[0].w = whh.data_ptr(); st.seg[0].ldw = H; st.seg[0].K = H
            st.pre = _ptr(pre, t0 * 8 * H + dr * 4 * H); st.ldpre = Lp * 8 * H
            st.c_prev = _ptr(cs[dr, slot_in]); st.ldc_prev = H
            st.h_out = _ptr(hs[dr, slot_out]); st.ldh = H
            st.h_out2 = _ptr(enc, t0 * E + dr * H); st.ldh2 = L * E
            st.c_out = _ptr(cs[dr, slot_out]); st.ldc_out = H
            st.gates_out = _ptr(gs[dr, t0]); st.ldg = 4 * H
            st.len = chars_len32.data_ptr(); st.t = t0
            ic = incs[dr]
            ic.seg_x[0] = sg * B * H
            ic.pre = sg * 8 * H; ic.c_prev = sg * B * H; ic.h_out = sg * B * H; ic.h_out2 = sg * E
            ic.c_out = sg * B * H; ic.gates_out = sg * B * 4 * H; ic.dt = sg
            if persist:
                st.xt = _ptr(ht[dr, slot_in]); st.ht_out = _ptr(ht[dr, slot_out]); st.ht_col0 = 0
                ic.xt = sg * H * Bp; ic.ht_out = sg * H * Bp
        self.mark("fwd.enc.convs")
        if persist:
            call("t2_lstm_seq_fwd_persist_pz", steps, incs, 2, S, self.persist_counters((B + 31) // 32), self.persist_flag(), _stream())
            # (a wait that timed out - sticky device flag - turns the encoder output into NaN: every later result of this forward,
            #  training or inference, carries it; the host raises at its next check_persistent_kernels)
            call("t2_guard_poison", self._persist_sync.data_ptr() + 4 * 256, enc, B * L * E, _stream())
        else:
            call("t2_lstm_seq_fwd", steps, incs, 2, S, _stream())
        ctx["enc_stash"] = dict(x3=x, pre=pre, hs=hs, cs=cs, gs=gs, B=B, L=L)
        return enc

    # ---- full teacher-forced forward -------------------------------------------------------------
    def controls_terms(self, controls, B):
        """Prosody controls (model/tacotron2.py:279-286, model/decoder.py:94-109): the same (B, controls_dim) vector is
        appended to the decoder-LSTM input and to the mel projection input at every frame, so its contribution is one
        per-utterance term for each: cterm (B, 4D) and cmel1 (B, M+1) (stop column zero)."""
        d, P = self.d, self.ps.P
        C = d.get("controls_dim", 0) if d.get("controls") else 0
        assert (controls is not None) == bool(C), \
            "Controls are enabled, but no control vector was passed to the model!" if C else \
            "Controls are disabled, but a control vector was passed to the model!"
        if not C:
            return None, None, None
        M, D = d["num_mels"], d["rnn_hidden_dim"]
        ctl = controls.to(self.dev, torch.float32).contiguous()
        assert tuple(ctl.shape) == (B, C), f"controls must be (B, {C})"
        cterm = self.buf("ctl.dec", B, 4 * D)
        gemm(ctl, P["decoder.lstm.weight_ih#controls"], cterm, B, 4 * D, C, C, C, 4 * D)
        cmel1 = self.buf("ctl.mel", B, M + 1, zero=True)
        gemm(ctl, P["decoder.mel_out.weight#controls"], cmel1, B, M, C, C, C, M + 1)
        return ctl, cterm, cmel1

    def forward_tf(self, chars_idx, chars_len, mel, mel_len, speaker_id=None, description_embeddings=None,
                   training=True, masks: Optional[dict] = None, save_for_backward=True, controls=None):
        """Returns (mels, mels_post, gates, alignments), ctx.  masks: oracle-convention dict (see oracle.tacotron2_ref)
        already on the device, with prenet/att/dec masks time-major; None entries = identity."""
        d, P, ps = self.d, self.ps.P, self.ps
        masks = masks or {}
        B, L = chars_idx.shape
        T, M = mel.shape[1], d["num_mels"]
        E, Pd, A, D, Ad = d["encoded_dim"], d["prenet_dim"], d["att_rnn_dim"], d["rnn_hidden_dim"], d["att_dim"]
        Ef = E + (128 if d.get("description_embeddings") else 0)
        F = d.get("loc_filters", 32)
        ctx: dict = dict(B=B, L=L, T=T)
        self.generation += 1          # any forward (grad-enabled or not) rewrites the shared workspaces
        ctx["generation"] = self.generation
        st = _stream()
        len32 = chars_len.to(torch.int32)
        mlen32 = mel_len.to(torch.int32)
        ctx["len32"], ctx["mlen32"], ctx["chars_idx"] = len32, mlen32, chars_idx

        self.begin_phase(backward=False)
        self.mark("start")
        # The prenet and the hoisted prenet part of the attention-RNN input projection depend only on the mel input: they run
        # on the side stream next to the encoder, whose BiLSTM recurrence is a chain of small latency-bound launches.
        main0, side0 = torch.cuda.current_stream(), self.side_stream()
        side0.wait_stream(main0)
        with torch.cuda.stream(side0):
            mel_tm = self.buf("mel_tm", T + 1, B, M)
            call("t2_mel_to_tm", mel, mel_tm, B, T, M, _stream())
            pd = masks.get("prenet_drop")
            p1 = self.buf("p1", T + 1, B, Pd)
            p2 = self.buf("p2", T + 1, B, Pd)
            R1 = (T + 1) * B
            gemm(mel_tm, P["prenet.0.weight"], p1, R1, Pd, M, M, M, Pd, relu=1, mulmask=pd[0] if pd else None, ldmask=Pd)
            gemm(p1, P["prenet.3.weight"], p2, R1, Pd, Pd, Pd, Pd, Pd, relu=1, mulmask=pd[1] if pd else None, ldmask=Pd)
            R = T * B
            pre_att = self.buf("pre_att", T, B, 4 * A)
            sp0 = self.span_begin()
            gemm(p2, P["decoder.att_rnn.weight_ih"], pre_att, R, 4 * A, Pd, Pd, Pd + Ef, 4 * A,
                 bias=P["decoder.att_rnn.bias_ih"], bias2=P["decoder.att_rnn.bias_hh"])
            self.span_end("fwd.dec.pre_att_gemm.side_stream", sp0)   # part of the decoder step; runs next to the encoder
        enc = self.encoder_fwd(chars_idx, len32, training, masks, ctx)
        self.mark("fwd.enc.bilstm")

        # conditioning (model/tacotron2.py:201-229)
        memory = self.buf("memory", B, L, Ef)
        desc = None
        if d.get("description_embeddings"):
            desc = self.buf("desc", B, 128)
            Dd = d["description_embeddings_dim"]
            gemm(description_embeddings, P["description_embeddings_linear.0.weight"], desc, B, 128, Dd, Dd, Dd, 128)
            call("t2_tanh_bias", desc, P["description_embeddings_linear.0.bias"], B, 128, st)
        spk32 = speaker_id.to(torch.int32) if d.get("speaker_tokens") else None
        call("t2_condition_fwd", enc, P["speaker_embedding.weight"] if d.get("speaker_tokens") else None, spk32, desc,
             memory, B, L, E, Ef, st)
        ctx.update(enc=enc, memory=memory, desc=desc, spk32=spk32, desc_in=description_embeddings)
        pmT = self.buf("pmT", B, Ad, L)   # processed memory, transposed: one batched GEMM W_att x memory[b]^T
        gemm(P["att_encoder.weight"], memory, pmT, Ad, L, Ef, Ef, Ef, L, batch=B, sA=0, sB=L * Ef, sC=Ad * L)

        self.mark("fwd.condition")
        main0.wait_stream(side0)      # prenet (model/tacotron2.py:255-258) and pre_att (side stream, above) are ready
        self.mark("fwd.dec.pre_att_gemm")
        # attention chain
        U = self.buf("U", Ad, 2, KL)
        call("t2_attn_fold_location", P["decoder.attention.location_dense.weight"],
             P["decoder.attention.location_conv.weight"], U, Ad, F, KL, st)
        xdec = self.buf("xdec", T + 1, B, A + Ef)
        zero_later(xdec[0])
        att_c = self.buf("att_c", T + 1, B, A)
        zero_later(att_c[0])
        cum = self.buf("cum", T + 1, B, L)
        zero_later(cum[0])
        xproj = self.buf("xproj", T + 1, B, D + Ef)
        zero_later(xproj[0])
        # x16-tiled copies of the recurrent inputs ([K/16][Bp][16] per slot): the step kernels' activation loads become
        # contiguous 1 KB blocks (include/tacotron2_amd.h, T2LstmStep.xt)
        Bp = (B + 15) // 16 * 16
        xdec_t = self.buf("xdec_t", T + 1, (A + Ef) // 16, Bp, 16, zero=(B != Bp))
        zero_later(xdec_t[0])
        dech_t = self.buf("dech_t", T + 1, D // 16, Bp, 16, zero=(B != Bp))
        zero_later(dech_t[0])
        gates_att = self.buf("gates_att", T, B, 4 * A) if save_for_backward else None
        # tanh terms of the energies, kept for the backward: [T][B][Ad][L4] floats, 2.7 GB per step at b=32, T=870, L=188 - sized for
        # 288 GB of HBM (a backward that recomputes them instead was built and measured in round 3: +1.0 ms per step,
        # profiles/r03_ab_tanh_recompute.txt)
        th = self.buf("th", T, B, Ad, (L + 3) // 4 * 4) if save_for_backward else None
        align = torch.empty(B, T, L, dtype=torch.float32, device=self.dev)
        e_part = self.buf("e_part", B, Ad // 16, L)
        # packed in the column order of the xdec row [att_h | ctx], so each step reads ONE contiguous activation segment
        wp_att = self.pack_fwd("att", [(P["decoder.att_rnn.weight_hh"], A, A),
                                       (_ptr(P["decoder.att_rnn.weight_ih"], Pd), Pd + Ef, Ef)], A)
        seq = make("T2AttnSeq", B=B, L=L, T=T, A=A, Ad=Ad, Ef=Ef, Kl=KL, wpacked=wp_att,
                   W_ih_ctx=_ptr(P["decoder.att_rnn.weight_ih"], Pd), ld_wih=Pd + Ef,
                   W_hh=P["decoder.att_rnn.weight_hh"], Wq=P["decoder.attention.query_layer.weight"], U=U,
                   v=P["decoder.attention.v.weight"], pre=pre_att, pmT=pmT, memory=memory, len=len32,
                   att_drop=masks.get("att_drop"), xdec=xdec, att_c=att_c, gates=gates_att, align=align, cum=cum, th=th,
                   xproj_ctx=_ptr(xproj, B * (D + Ef) + D), ld_xproj=D + Ef, e_part=e_part, xdec_t=xdec_t)
        # decoder-LSTM chain operands (prepared before the pipeline below)
        pre_dec = self.buf("pre_dec", T, B, 4 * D)
        dec_c = self.buf("dec_c", T + 1, B, D)
        zero_later(dec_c[0])
        gates_dec = self.buf("gates_dec", T, B, 4 * D) if save_for_backward else None
        dd = masks.get("dec_drop")
        ldp = D + Ef
        wp_dec = self.pack_fwd("dec", [(P["decoder.lstm.weight_hh"], D, D)], D)
        # Software pipeline over chunks of CH frames.  In teacher-forced mode the attention chain never reads the decoder
        # LSTM (model/decoder.py:70-101): the attention chain of chunk i runs on the main stream while the decoder-LSTM chain of
        # chunk i-1 - its hoisted input-projection GEMM, then the recurrence - runs on the side stream.  (Round 1 co-scheduled the
        # decoder steps inside the attention-energies launches instead; that stretched every frame of the critical chain and was
        # removed in round 4: profiles/r02_ab_fwd_dec_chain.txt.)
        CH = self.chunk

        def dec_chunk(c0, c1):
            stp = make("T2LstmStep", B=B, H=D, nseg=1, wpacked=wp_dec, pre=_ptr(pre_dec, c0 * B * 4 * D), ldpre=4 * D,
                       c_prev=_ptr(dec_c, c0 * B * D), ldc_prev=D,
                       drop=_ptr(dd, c0 * B * D) if dd is not None else None, lddrop=D,
                       h_out=_ptr(xproj, (c0 + 1) * B * ldp), ldh=ldp, c_out=_ptr(dec_c, (c0 + 1) * B * D), ldc_out=D,
                       gates_out=_ptr(gates_dec, c0 * B * 4 * D) if gates_dec is not None else None, ldg=4 * D,
                       xt=_ptr(dech_t, c0 * D * Bp), ht_out=_ptr(dech_t, (c0 + 1) * D * Bp), ht_col0=0)
            stp.seg[0].x = _ptr(xproj, c0 * B * ldp); stp.seg[0].ldx = ldp
            stp.seg[0].w = P["decoder.lstm.weight_hh"].data_ptr(); stp.seg[0].ldw = D; stp.seg[0].K = D
            inc = make("T2LstmStride", pre=B * 4 * D, c_prev=B * D, drop=B * D, h_out=B * ldp, c_out=B * D,
                       gates_out=B * 4 * D, dt=0, xt=D * Bp, ht_out=D * Bp)
            inc.seg_x[0] = B * ldp
            return stp, inc

        ctl, cterm, cmel1 = self.controls_terms(controls, B)

        def pre_dec_gemm(c0, c1):
            if cterm is not None:     # per-utterance controls term first, the projection accumulates on top
                pre_dec[c0:c1].copy_(cterm.unsqueeze(0).expand(c1 - c0, B, 4 * D))
            # the short chunks at the end of the ramp leave the chip under-filled (8 frames x 32 rows: 64 tiles of 128 x 128) and
            # sit on the exposed tail of the forward: K slices accumulate into the pre-filled block with atomics
            tiles = (((c1 - c0) * B + 127) // 128) * ((4 * D + 127) // 128)
            sk = max(1, min(4, 256 // max(tiles, 1))) if self.splitk_small_chunks else 1
            if sk > 1 and cterm is None and not _PREZEROED.pop(_ptr(pre_dec, c0 * B * 4 * D), False):
                zero_later(pre_dec[c0:c1])
            gemm(_ptr(xdec, (c0 + 1) * B * (A + Ef)), P["decoder.lstm.weight_ih"], _ptr(pre_dec, c0 * B * 4 * D), (c1 - c0) * B,
                 4 * D, A + Ef, A + Ef, A + Ef, 4 * D, bias=P["decoder.lstm.bias_ih"], bias2=P["decoder.lstm.bias_hh"],
                 accumulate=2 if sk > 1 else (1 if cterm is not None else 0), splitk=sk)

        main, side = torch.cuda.current_stream(), self.side_stream()
        if cterm is None and self.splitk_small_chunks:
            # the short chunks of the ramp run their hoisted GEMM split-K into a cleared block: cleared here, with everything else
            c0_ = 0
            for n_ in _chunk_sizes(T, self.chunk, ramp_at_end=self.ramp_chunks):
                if max(1, min(4, 256 // max(((n_ * B + 127) // 128) * ((4 * D + 127) // 128), 1))) > 1:
                    zero_later(pre_dec[c0_:c0_ + n_]); _PREZEROED[_ptr(pre_dec, c0_ * B * 4 * D)] = True
                c0_ += n_
            flush_zeros()              # (on the main stream, in front of the side stream's wait below)
        side.wait_stream(main)
        # Pipeline chunks; the LAST ones shrink (CH/2, CH/4, CH/8, CH/8): what follows the attention chain's end on the side stream
        # (the decoder-LSTM frames of the final chunk) is exposed time, proportional to that chunk's length.
        sizes = _chunk_sizes(T, CH, ramp_at_end=self.ramp_chunks)
        chunks, c0 = [], 0
        for n in sizes:
            chunks.append((c0, c0 + n)); c0 += n
        persist = B <= 64 and self.dec_chain == "persistent" and D // 4 <= 256 and self.persist_resident(D, B)
        ctx["persist"] = persist
        sync = self.persist_sync() if persist else None
        for i, (c0, c1) in enumerate(chunks):
            seq.t_begin, seq.t_end = c0, c1
            call("t2_attn_seq_fwd", seq, st)
            if persist and not self.persist_gemm_side:
                pre_dec_gemm(c0, c1)
            ev = main.record_event()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                if not persist or self.persist_gemm_side:
                    SHARE_CU[0] = self.share_cu if persist else 0
                    pre_dec_gemm(c0, c1)
                    SHARE_CU[0] = 0
                stp, inc = dec_chunk(c0, c1)
                if persist:
                    # The decoder-LSTM chain of a chunk as ONE persistent, weight-stationary launch (t2_lstm_seq_fwd_persist):
                    # W_hh stays in LDS, the workgroups exchange h through the tiled stash
                    call("t2_lstm_seq_fwd_persist_pz", stp, inc, 1, c1 - c0, self.persist_counters((B + 31) // 32), self.persist_flag(),
                         side.cuda_stream)
                else:
                    call("t2_lstm_seq_fwd", stp, inc, 1, c1 - c0, side.cuda_stream)
        self.mark("fwd.dec.attn_chain")
        main.wait_stream(side)
        self.mark("fwd.dec.lstm_chain_tail")

        # mel + stop projection over all frames: [mel_out.weight ; gate.weight] is one (M+1, D+Ef) matrix
        wproj = ps.cat_view("decoder.mel_out.weight", M + 1, D + Ef)
        bproj = ps.cat_view("decoder.mel_out.bias", M + 1, 0)
        proj = self.buf("proj", T, B, M + 1)
        if cmel1 is not None:
            proj.copy_(cmel1.unsqueeze(0).expand(T, B, M + 1))
        gemm(_ptr(xproj, B * ldp), wproj, proj, R, M + 1, ldp, ldp, ldp, M + 1, bias=bproj,
             accumulate=1 if cmel1 is not None else 0)
        if persist:
            # A persistent launch that gave up on an inter-workgroup wait (sticky device flag) must not pass unnoticed: with the
            # flag set the projection - and with it every output, the loss and every gradient of this step - becomes NaN, and
            # t2_adam_step skips a step with a non-finite gradient norm.  No host synchronisation here; the host raises where it
            # reads the loss anyway (check_persistent_kernels).
            call("t2_guard_poison", self._persist_sync.data_ptr() + 4 * 256, proj, R * (M + 1), st)
        self.mark("fwd.dec.proj_gemm")
        mels = torch.empty(B, T, M, dtype=torch.float32, device=self.dev)
        gates = torch.empty(B, T, 1, dtype=torch.float32, device=self.dev)
        post_in = self.buf("post.x0", B, T + 4, M)
        call("t2_finalize_fwd", proj, M + 1, mlen32, mels, gates, post_in, B, T, M, st)

        # postnet (model/postnet.py) + residual + output masks (model/tacotron2.py:331-345)
        Pn = d["postnet_dim"]
        chans = [M, Pn, Pn, Pn, Pn, M]
        post_drop = masks.get("post_drop")
        x = post_in
        post = torch.empty(B, T, M, dtype=torch.float32, device=self.dev)
        for li in range(5):
            last = li == 4
            x = self.conv_bn_fwd(f"post.conv{li}", x, P[f"postnet.postnet.{4 * li}.weight"], None,
                                 f"postnet.postnet.{4 * li + 1}", B, T, chans[li], chans[li + 1], 0 if last else 2,
                                 post_drop[li] if post_drop is not None else None, training, ctx,
                                 y=post if last else None, Lp_y=T if last else None, pad_y=0 if last else 2,
                                 res=post_in if last else None, Lp_res=T + 4, pad_res=2,
                                 length=mlen32 if last else None, fill=0.0)
        self.mark("fwd.postnet")
        ctx.update(controls=ctl, pmT=pmT, mel_tm=mel_tm, p1=p1, p2=p2, pd=pd, pre_att=pre_att, U=U, xdec=xdec, att_c=att_c, cum=cum,
                   xproj=xproj, gates_att=gates_att, th=th, align=align, pre_dec=pre_dec, dec_c=dec_c,
                   gates_dec=gates_dec, proj=proj, post_in=post_in, masks=masks, training=training)
        return (mels, post, gates, align), ctx

    # =============================================================================================
    # backward
    # =============================================================================================
    def _wgrad(self, dY, ldy, X, ldx, Cgrad, ldc, Mout, Nin, R):
        """Cgrad[Mout, Nin] += dY[R, Mout]^T @ X[R, Nin]  (split-K, fp32 atomics into the zero-initialised grad buffer)."""
        gemm(dY, X, Cgrad, Mout, Nin, R, ldy, ldx, ldc, a_k=0, b_k=0, accumulate=2, splitk=splitk_for(Mout, Nin, R))

    def conv_bn_bwd(self, tag, ctx, dy, Lp_dy, pad_dy, w, gw, gbias, bn_prefix, B, L, Ci, Co, act, training, need_dx=True,
                    defer=None, wgrad_stream=None):
        """Backward of conv_bn_fwd.  dy: gradient w.r.t. the layer output.  Returns dX in shifted rows (B*(L+4), Ci)."""
        P, G = self.ps.P, self.ps.G
        c = ctx[tag]
        Lp = L + 4
        st = _stream()
        draw = self.buf(f"{tag}.draw", B, Lp, Co)
        sums = self.bn_sums(tag, backward=True)
        bn = make("T2Bn", B=B, L=L, C=Co, x=c["raw"], Lp_x=Lp, gamma=P[bn_prefix + ".weight"], beta=P[bn_prefix + ".bias"],
                  training=1 if training else 0, momentum=0.1, eps=1e-5, sums=sums, sums_prezeroed=1, mean=c["mean"], invstd=c["invstd"],
                  act=act, drop=c["drop"], dy=dy, Lp_dy=Lp_dy, pad_dy=pad_dy, dx=draw, Lp_dx=Lp, pad_dx=2,
                  dgamma=G[bn_prefix + ".weight"], dbeta=G[bn_prefix + ".bias"])
        if training and self.sync_bn_group is not None:
            # the batch-statistics terms of the BN backward (sum dz, sum dz * xhat) are sums over the GLOBAL batch too
            import torch.distributed as dist
            bn.phase = 1
            call("t2_bn_bwd", bn, st)
            dist.all_reduce(sums, group=self.sync_bn_group)
            bn.phase = 2; bn.grad_share = 1.0 / dist.get_world_size(self.sync_bn_group)
        call("t2_bn_bwd", bn, st)
        R = B * Lp - 4
        if gbias is not None:
            call("t2_colsum", draw, Co, B * Lp, Co, gbias, st)
        def wgrad():     # not on the critical path: with `defer` it is run later, on whatever stream is current then
            dwp = self.buf(f"{tag}.dwp", Co, 5 * Ci, zero=True)
            self._wgrad(_ptr(draw, 2 * Co), Co, c["x_pad"], Ci, dwp, 5 * Ci, Co, 5 * Ci, R)
            call("t2_unpack_conv_wgrad", dwp, gw, Co, Ci, 5, _stream())
        if defer is not None:
            defer.append(wgrad)
        elif wgrad_stream is not None:      # at once, but on another stream (behind everything enqueued here so far)
            ev = torch.cuda.current_stream().record_event()
            with torch.cuda.stream(wgrad_stream):
                wgrad_stream.wait_event(ev)
                wgrad()
        else:
            wgrad()
        if not need_dx:
            return None
        wf = self.buf(f"{tag}.wf", Ci, 5 * Co)
        call("t2_pack_conv_weight", w, wf, Co, Ci, 5, 1, st)
        dx = self.buf(f"{tag}.dx", B * Lp, Ci)
        gemm_fill(draw, wf, dx, R, Ci, 5 * Co, Co, 5 * Co, Ci)
        return dx

    def backward_tf(self, ctx, d_post, dproj):
        """d_post (B,T,M): gradient w.r.t. mels_post (masked positions zero).  dproj [T][B][M+1]: gradient w.r.t. the
        decoder projection from the mel / residual / gate terms.  Accumulates into ps.grad (caller zeroes it)."""
        d, P, G, ps = self.d, self.ps.P, self.ps.G, self.ps
        if ctx.get("generation") != self.generation:
            raise RuntimeError("backward of a stale forward: the activation stashes of this forward were overwritten by a later "
                               "grad-enabled forward of the same model (one live teacher-forced graph per model; INTEGRATION.md)")
        B, L, T = ctx["B"], ctx["L"], ctx["T"]
        M, E, Pd, A, D, Ad = d["num_mels"], d["encoded_dim"], d["prenet_dim"], d["att_rnn_dim"], d["rnn_hidden_dim"], d["att_dim"]
        Ef = E + (128 if d.get("description_embeddings") else 0)
        F = d.get("loc_filters", 32)
        H = E // 2
        Pn = d["postnet_dim"]
        st = _stream()
        training = ctx["training"]
        masks = ctx["masks"]
        R, R1 = T * B, (T + 1) * B
        ldp, ldx = D + Ef, A + Ef
        self.begin_phase(backward=True)
        # accumulators of the deferred weight-gradient GEMMs (run later, mostly on the side stream) and the split-K outputs of the
        # encoder's data-gradient GEMMs: cleared here, in the backward's first launch, not one launch each
        pchans = [M, Pn, Pn, Pn, Pn, M]
        for li in range(5):
            self.prezero(f"post.conv{li}.dwp", pchans[li + 1], 5 * pchans[li])
        Lp_e = L + 4
        for li in range(3):
            self.prezero(f"enc.conv{li}.dwp", E, 5 * E)
            if fill_splits(B * Lp_e - 4, E, 5 * E):
                dxe = self.buf(f"enc.conv{li}.dx", B * Lp_e, E)
                zero_later(dxe.view(-1)[:(B * Lp_e - 4) * E]); _PREZEROED[dxe.data_ptr()] = True
        if fill_splits(B * Lp_e - 4, E, 8 * H):
            dxe = self.buf("enc.dx3", B * Lp_e, E)
            zero_later(dxe.view(-1)[:(B * Lp_e - 4) * E]); _PREZEROED[dxe.data_ptr()] = True

        # ---- postnet --------------------------------------------------------------------------------
        chans = [M, Pn, Pn, Pn, Pn, M]
        dy, Lp_dy = d_post, T
        # the five weight-gradient GEMMs of the postnet (1.5 ms) leave the critical path: they run on the side stream between the
        # chunks of the backward frame loop, where the decoder-LSTM chain has slack
        post_wgrads = [] if self.defer_wgrads else None
        for li in range(4, -1, -1):
            dy = self.conv_bn_bwd(f"post.conv{li}", ctx, dy, Lp_dy, 0, P[f"postnet.postnet.{4 * li}.weight"],
                                  G[f"postnet.postnet.{4 * li}.weight"], None, f"postnet.postnet.{4 * li + 1}", B, T,
                                  chans[li], chans[li + 1], 0 if li == 4 else 2, training, defer=post_wgrads)
            Lp_dy = T + 4
        post_wgrads = post_wgrads or []
        call("t2_finalize_bwd", dy, dproj, B, T, M, st)
        self.mark("bwd.postnet")

        # ---- mel/stop projection ----------------------------------------------------------------------
        xproj, xdec = ctx["xproj"], ctx["xdec"]
        wproj = ps.cat_view("decoder.mel_out.weight", M + 1, ldp)
        dxproj = self.buf("dxproj", T, B, ldp)
        gemm(dproj, wproj, dxproj, R, ldp, M + 1, M + 1, ldp, ldp, a_k=1, b_k=0)
        ctl = ctx.get("controls")

        def proj_wgrads():       # weight / bias gradients of the projection: off the critical path (deferred with the postnet's)
            self._wgrad(dproj, M + 1, _ptr(xproj, B * ldp), ldp, ps.cat_view("decoder.mel_out.weight", M + 1, ldp, grad=True),
                        ldp, M + 1, ldp, R)
            call("t2_colsum", dproj, M + 1, R, M + 1, ps.cat_view("decoder.mel_out.bias", M + 1, 0, grad=True), _stream())
            if ctl is not None:       # controls columns: sum the gradient over frames per utterance, then (M, B) x (B, C)
                C = ctl.shape[1]
                s_proj = self.buf("ctl.dproj_sum", B, M + 1, zero=True)
                call("t2_colsum", dproj, B * (M + 1), T, B * (M + 1), s_proj, _stream())
                self._wgrad(s_proj, M + 1, ctl, C, G["decoder.mel_out.weight#controls"], C, M, C, B)
        if self.defer_wgrads:
            post_wgrads.append(proj_wgrads)
        else:
            proj_wgrads()

        # ---- both recurrences, back-propagation through time, as a two-stream pipeline over chunks of frames -----------
        # side stream: decoder-LSTM BPTT of chunk k (1 launch / frame) + the GEMM that turns its gate gradients into
        #              d[att_h, ctx] for those frames;   main stream: attention-chain BPTT of chunk k (4 launches / frame).
        # The attention chain of a frame only needs the decoder chain's gradient of the SAME frame, so chunk k+1 of the
        # decoder chain overlaps chunk k of the attention chain; the decoder weight-gradient GEMMs overlap the tail.
        dgd = self.buf("dgd", T + 1, B, 4 * D)
        zero_later(dgd[T])
        # x16-tiled copies of the gate gradients (A operands of the per-frame backward products; T2LstmBwdStep.dgt_next)
        Bp = (B + 15) // 16 * 16
        dgd_t = self.buf("dgd_t", T + 1, 4 * D // 16, Bp, 16)
        zero_later(dgd_t[T])
        Zt = self.buf("Zatt_t", T + 1, 4 * A // 16, Bp, 16)
        zero_later(Zt[T])
        dc_dec = self.buf("dc_dec", B, D, zero=True)
        dd = masks.get("dec_drop")
        wtp_dec = self.pack_bwd("dec.t", P["decoder.lstm.weight_hh"], D, 4 * D, D)
        dxdec = self.buf("dxdec", T, B, ldx)
        ldz = 4 * A + Ad
        Z = self.buf("Zatt", T + 1, B, ldz)       # Z[s][b] = [dgates_s | dq_{s-1}]
        zero_later(Z[T, :, :4 * A])
        dga = Z                                   # dgates_t = Z[t][:, :4A]   (row stride ldz)
        dctx_tot = self.buf("dctx_tot", T, B, Ef)
        dpmT = self.buf("dpmT", B, Ad, L, zero=True)
        dv_part = self.buf("dv_part", B, Ad, zero=True)
        dU_part = self.buf("dU_part", B, Ad * 2 * KL, zero=True)
        dc_att = self.buf("dc_att", B, A, zero=True)
        Gc = self.buf("Gcum", 2, B, L)
        de = self.buf("de", B, L)
        din_part = self.buf("din_part", B, Ad // 16, 2, L)
        wtp_ctx = self.pack_bwd("att.ctx.t", _ptr(P["decoder.att_rnn.weight_ih"], Pd), Pd + Ef, 4 * A, Ef)
        wtp_h = self.pack_bwd("att.h.t", P["decoder.att_rnn.weight_hh"], A, 4 * A, A)
        wtp_q = self.pack_bwd("att.q.t", P["decoder.attention.query_layer.weight"], A, Ad, A)
        dh_rec = self.buf("dh_rec", B, A)
        sb = make("T2AttnSeqBwd", B=B, L=L, T=T, A=A, Ad=Ad, Ef=Ef, Kl=KL, wtp_ctx=wtp_ctx, wtp_h=wtp_h, wtp_q=wtp_q,
                  dh_rec=dh_rec,
                  W_ih_ctx=_ptr(P["decoder.att_rnn.weight_ih"], Pd), ld_wih=Pd + Ef, W_hh=P["decoder.att_rnn.weight_hh"],
                  Wq=P["decoder.attention.query_layer.weight"], U=ctx["U"], v=P["decoder.attention.v.weight"],
                  memory=ctx["memory"], xdec=xdec, att_c=ctx["att_c"], gates=ctx["gates_att"], align=ctx["align"],
                  cum=ctx["cum"], th=ctx["th"], att_drop=masks.get("att_drop"),
                  dh_ext=dxdec, ld_dh=ldx, dctx_ext1=_ptr(dxdec, A), ld_dc1=ldx, dctx_ext2=_ptr(dxproj, D), ld_dc2=ldp,
                  dgates=Z, dctx_tot=dctx_tot, dq=None, dpmT=dpmT, dv_part=dv_part, dU_part=dU_part,
                  dc=dc_att, G=Gc, de=de, din_part=din_part, dgates_t=Zt, clk=getattr(self, "clk_bwd", None),
                  ws_bd=self.buf("attn.ws_bd", Ad // 16 * 16896))
        self.mark("bwd.dec.proj")
        main, side = torch.cuda.current_stream(), self.side_stream()
        side.wait_stream(main)
        CH = self.chunk_bwd

        def dec_bwd_chunk(hi, lo):
            s = make("T2LstmBwdStep", B=B, H=D, N4=4 * D, dg_next=_ptr(dgd, hi * B * 4 * D), lddg=4 * D,
                     W=P["decoder.lstm.weight_hh"], ldw=D, wtpacked=wtp_dec, ncols=D, epi=1,
                     ext1=_ptr(dxproj, (hi - 1) * B * ldp), ldx1=ldp,
                     drop=_ptr(dd, (hi - 1) * B * D) if dd is not None else None, lddrop=D,
                     gates=_ptr(ctx["gates_dec"], (hi - 1) * B * 4 * D), ldgs=4 * D,
                     c_prev=_ptr(ctx["dec_c"], (hi - 1) * B * D), ldcp=D, c_cur=_ptr(ctx["dec_c"], hi * B * D), ldcc=D,
                     dc=dc_dec, lddc=D, dg_out=_ptr(dgd, (hi - 1) * B * 4 * D), ldgo=4 * D,
                     dgt_next=_ptr(dgd_t, hi * Bp * 4 * D), dgt_out=_ptr(dgd_t, (hi - 1) * Bp * 4 * D),
                     off_chain=1 if self.bptt_off_chain else 0)
            inc = make("T2LstmBwdStride", dg=-B * 4 * D, ext1=-B * ldp, drop=-B * D, gates=-B * 4 * D, c_prev=-B * D,
                       c_cur=-B * D, dt=0, dgt=-Bp * 4 * D)
            return s, inc

        def dxdec_gemm(hi, lo, share_cu=0):
            SHARE_CU[0] = share_cu
            gemm(_ptr(dgd, lo * B * 4 * D), P["decoder.lstm.weight_ih"], _ptr(dxdec, lo * B * ldx), (hi - lo) * B, ldx, 4 * D,
                 4 * D, ldx, ldx, a_k=1, b_k=0)
            SHARE_CU[0] = 0

        def dec_wgrads_chunk(hi, lo):    # decoder-LSTM weight gradients over frames [lo, hi) (accumulating)
            n = (hi - lo) * B
            g0 = _ptr(dgd, lo * B * 4 * D)
            self._wgrad(g0, 4 * D, _ptr(xdec, (lo + 1) * B * ldx), ldx, G["decoder.lstm.weight_ih"], ldx, 4 * D, ldx, n)
            self._wgrad(g0, 4 * D, _ptr(xproj, lo * B * ldp), ldp, G["decoder.lstm.weight_hh"], D, 4 * D, D, n)

        def dec_wgrads():   # decoder-LSTM weight gradients, on the side stream next to the attention chain's tail
            with torch.cuda.stream(side):
                if not self.chunk_att_wgrads:
                    SHARE_CU[0] = self.share_cu
                    dec_wgrads_chunk(T, 0)
                    SHARE_CU[0] = 0
                db = self.buf("db_dec", 4 * D, zero=True)                      # both biases see the same gate gradients
                call("t2_colsum", dgd, 4 * D, R, 4 * D, db, side.cuda_stream)
                for nm in ("decoder.lstm.bias_ih", "decoder.lstm.bias_hh"):    # t2_colsum over ONE row = accumulate
                    call("t2_colsum", db, 4 * D, 1, 4 * D, G[nm], side.cuda_stream)
                if ctl is not None:
                    s_dgd = self.buf("ctl.dgd_sum", B, 4 * D, zero=True)
                    call("t2_colsum", dgd, B * 4 * D, T, B * 4 * D, s_dgd, side.cuda_stream)
                    self._wgrad(s_dgd, 4 * D, ctl, ctl.shape[1], G["decoder.lstm.weight_ih#controls"], ctl.shape[1], 4 * D,
                                ctl.shape[1], B)

        # (time-descending) the FIRST chunks are short: the attention chain cannot start before the decoder-LSTM BPTT of its first
        # chunk and that chunk's GEMM are done on the side stream
        chunks, hi = [], T
        for n in reversed(_chunk_sizes(T, CH, ramp_at_end=self.ramp_chunks)):
            chunks.append((hi, hi - n)); hi -= n
        # Two-stream pipeline: decoder chain of chunk k+1 on the side stream next to the attention chain of chunk k.  Hosting the
        # decoder BPTT steps inside attention launches instead measured slower every way (inside the ds launch: round 1,
        # profiles/r01_sweep_bwd_chunk_co.txt; as a second operand block of the cell-backward launch: 75.3 against 71.9 ms
        # per step, profiles/r02_ab_bwd_schedule.txt; as a third descriptor of the products launch - a kernel of the same kind -
        # 71.0 against 66.6 ms, profiles/r03_ab_bwd_host.txt): as its own launch the step overlaps the latency-bound dw / ds /
        # cell-backward launches of the chain, hosted it doubles the load of the CUs that carry it.
        gWih = G["decoder.att_rnn.weight_ih"]

        def att_wgrads(hi, lo):      # weight gradients of the attention chain over frames [lo, hi) (accumulating: split-K atomics)
            n = (hi - lo) * B
            z0 = _ptr(Z, lo * B * ldz)
            self._wgrad(z0, ldz, _ptr(ctx["p2"], lo * B * Pd), Pd, gWih, Pd + Ef, 4 * A, Pd, n)
            self._wgrad(z0, ldz, _ptr(xdec, lo * B * ldx + A), ldx, _ptr(gWih, Pd), Pd + Ef, 4 * A, Ef, n)
            self._wgrad(z0, ldz, _ptr(xdec, lo * B * ldx), ldx, G["decoder.att_rnn.weight_hh"], A, 4 * A, A, n)
            self._wgrad(_ptr(Z, (lo + 1) * B * ldz + 4 * A), ldz, _ptr(xdec, (lo + 1) * B * ldx), ldx,
                        G["decoder.attention.query_layer.weight"], A, Ad, A, n)
        # Weight gradients along the pipeline: frames are handed to the weight-gradient GEMMs in groups of `wgrad_group` chunks
        # (a longer K per call keeps the split-K GEMMs efficient); ranges are contiguous and time-descending, so a group is one
        # [lo, hi) range.  dec_*: gate gradients of the decoder-LSTM BPTT (this stream's own output: no wait);  att_*: of the
        # attention chain (main stream: wait for the event recorded behind the group's last chunk).
        WG = max(1, int(self.wgrad_group))
        dec_grp, att_done, att_grp = None, [], None        # [hi, lo, n]; [(hi, lo, event)]; [hi, lo, n, event]
        for ci_, (hi, lo) in enumerate(chunks):
            with torch.cuda.stream(side):
                s, inc = dec_bwd_chunk(hi, lo)
                call("t2_lstm_seq_bwd", s, inc, 1, hi - lo, side.cuda_stream)
                dxdec_gemm(hi, lo, self.share_cu)
                ev = side.record_event()
                # behind the event (the attention chain does not wait for any of this)
                SHARE_CU[0] = self.share_cu
                if self.chunk_att_wgrads:
                    dec_grp = [hi, lo, 1] if dec_grp is None else [dec_grp[0], lo, dec_grp[2] + 1]
                    if dec_grp[2] >= WG:
                        dec_wgrads_chunk(dec_grp[0], dec_grp[1]); dec_grp = None
                if post_wgrads and ci_ >= 4:
                    post_wgrads.pop(0)()
                if self.chunk_att_wgrads and len(att_done) >= 2:      # chunks the main stream finished two chunks ago
                    h2, l2, e2 = att_done.pop(0)
                    att_grp = [h2, l2, 1, e2] if att_grp is None else [att_grp[0], l2, att_grp[2] + 1, e2]
                    if att_grp[2] >= WG:
                        side.wait_event(att_grp[3])
                        att_wgrads(att_grp[0], att_grp[1]); att_grp = None
                SHARE_CU[0] = 0
            main.wait_event(ev)
            sb.t_hi, sb.t_lo = hi, lo
            call("t2_attn_seq_bwd", sb, st)
            if self.chunk_att_wgrads:
                att_done.append((hi, lo, main.record_event()))
        with torch.cuda.stream(side):
            while post_wgrads:                      # (short sequences: fewer chunks than deferred GEMMs)
                post_wgrads.pop(0)()
            SHARE_CU[0] = self.share_cu
            if dec_grp is not None:
                dec_wgrads_chunk(dec_grp[0], dec_grp[1])
            for h2, l2, e2 in att_done:
                att_grp = [h2, l2, 1, e2] if att_grp is None else [att_grp[0], l2, att_grp[2] + 1, e2]
            if att_grp is not None:
                side.wait_event(att_grp[3])
                att_wgrads(att_grp[0], att_grp[1])
            SHARE_CU[0] = 0
        dec_wgrads()
        self.mark("bwd.dec.chains")

        # gradient w.r.t. the encoder memory: context path (batched over samples) + processed-memory path.  It heads the
        # critical branch (conditioning -> encoder BiLSTM recurrence -> encoder convolutions), so it goes first.
        dmem = self.buf("dmem", B, L, Ef)
        gemm(ctx["align"], dctx_tot, dmem, L, Ef, T, L, B * Ef, Ef, a_k=0, b_k=0, batch=B, sA=T * L, sB=Ef, sC=L * Ef)
        Watt = P["att_encoder.weight"]
        gemm(dpmT, Watt, dmem, L, Ef, Ad, L, Ef, Ef, a_k=0, b_k=0, accumulate=1, batch=B, sA=Ad * L, sB=0, sC=L * Ef)

        # Independent branch on the side stream: weight gradients of the attention chain (large GEMMs over all frames) and the
        # prenet backward.  The main stream meanwhile walks the encoder BiLSTM backward recurrence, a chain of small
        # latency-bound launches that fits next to the GEMM workgroups.
        side.wait_stream(main)
        with torch.cuda.stream(side):
            sst = side.cuda_stream
            if not self.chunk_att_wgrads:
                att_wgrads(T, 0)
            db = self.buf("db_att", 4 * A, zero=True)
            call("t2_colsum", dga, ldz, R, 4 * A, db, sst)
            for nm in ("decoder.att_rnn.bias_ih", "decoder.att_rnn.bias_hh"):
                call("t2_colsum", db, 4 * A, 1, 4 * A, G[nm], sst)
            call("t2_colsum", dv_part, Ad, B, Ad, G["decoder.attention.v.weight"], sst)
            dU = self.buf("dU", Ad, 2 * KL, zero=True)
            call("t2_colsum", dU_part, Ad * 2 * KL, B, Ad * 2 * KL, dU, sst)
            Wd, Wc = P["decoder.attention.location_dense.weight"], P["decoder.attention.location_conv.weight"]
            gemm(dU, Wc, G["decoder.attention.location_dense.weight"], Ad, F, 2 * KL, 2 * KL, 2 * KL, F, accumulate=1)
            gemm(Wd, dU, G["decoder.attention.location_conv.weight"], F, 2 * KL, Ad, F, 2 * KL, 2 * KL, a_k=0, b_k=0,
                 accumulate=1)
            gemm(dpmT, ctx["memory"], G["att_encoder.weight"], Ad, Ef, L, L, Ef, Ef, a_k=1, b_k=0, accumulate=2, batch=B,
                 sA=Ad * L, sB=L * Ef, sC=0)
            # ---- prenet ----
            dp2 = self.buf("dp2", T + 1, B, Pd)
            zero_later(dp2[T])
            gemm(dga, P["decoder.att_rnn.weight_ih"], dp2, R, Pd, 4 * A, ldz, Pd + Ef, Pd, a_k=1, b_k=0)
            pd = ctx["pd"]
            g2 = self.buf("g2", T + 1, B, Pd)
            call("t2_relu_mask_bwd", dp2, ctx["p2"], pd[1] if pd else None, g2, R1 * Pd, sst)
            self._wgrad(g2, Pd, ctx["p1"], Pd, G["prenet.3.weight"], Pd, Pd, Pd, R1)
            dp1 = self.buf("dp1", T + 1, B, Pd)
            gemm(g2, P["prenet.3.weight"], dp1, R1, Pd, Pd, Pd, Pd, Pd, a_k=1, b_k=0)
            g1 = self.buf("g1", T + 1, B, Pd)
            call("t2_relu_mask_bwd", dp1, ctx["p1"], pd[0] if pd else None, g1, R1 * Pd, sst)
            self._wgrad(g1, Pd, ctx["mel_tm"], M, G["prenet.0.weight"], M, Pd, M, R1)
            # Every gradient from `prenet.0.weight` to the end of the flat buffer (prenet, att_encoder, both decoder cells,
            # attention, projection, postnet, controls columns) is now enqueued on THIS stream; what is left - encoder, speaker
            # table, description linear - follows on the main stream.  A data-parallel trainer starts the all-reduce of that
            # tail here, behind this stream, so that it runs next to the encoder backward (Trainer.overlap_allreduce).
            if self.grad_tail_hook is not None:
                self.grad_tail_hook()
        self.mark("bwd.dec.attn_gemms")

        # ---- conditioning ---------------------------------------------------------------------------------------
        denc = self.buf("denc", B, L, E)
        ddesc = self.buf("ddesc", B, 128, zero=True) if d.get("description_embeddings") else None
        call("t2_condition_bwd", dmem, ctx["memory"], ctx["spk32"], denc,
             G["speaker_embedding.weight"] if d.get("speaker_tokens") else None, ddesc, B, L, E, Ef, st)
        if ddesc is not None:
            Dd = d["description_embeddings_dim"]
            dlin = self.buf("dlin", B, 128)
            call("t2_tanh_bwd", ddesc, ctx["desc"], dlin, B * 128, st)
            self._wgrad(dlin, 128, ctx["desc_in"], Dd, G["description_embeddings_linear.0.weight"], Dd, 128, Dd, B)
            call("t2_colsum", dlin, 128, B, 128, G["description_embeddings_linear.0.bias"], st)

        # ---- encoder BiLSTM ---------------------------------------------------------------------------------------
        e = ctx["enc_stash"]
        S, Lp = L, L + 4
        hs, cs, gs = e["hs"], e["cs"], e["gs"]
        dgt = self.buf("enc.dgt", 2, S + 1, B, 4 * H)   # dir 0: dgates_t at slot t (zero slot S); dir 1: at slot t+1 (zero slot 0)
        zero_later(dgt[0, S]); zero_later(dgt[1, 0])
        dpre = self.buf("enc.dpre", B * Lp, 8 * H, zero=True)
        dc_enc = self.buf("enc.dc", 2, B, H, zero=True)
        steps = (_lib.S["T2LstmBwdStep"] * 2)()
        incs = (_lib.S["T2LstmBwdStride"] * 2)()
        len32 = ctx["len32"]
        for dr in range(2):
            sg = -1 if dr == 0 else 1                 # BPTT runs against the forward processing order
            t0 = S - 1 if dr == 0 else 0
            whh = P["encoder.lstm.weight_hh_l0" + ("" if dr == 0 else "_reverse")]
            sp = steps[dr]
            sp.B, sp.H, sp.N4, sp.ncols, sp.epi = B, H, 4 * H, H, 1
            sp.W = whh.data_ptr(); sp.ldw = H
            sp.wtpacked = self.pack_bwd(f"enc.whh{dr}.t", whh, H, 4 * H, H).data_ptr()
            sp.ext1 = _ptr(denc, t0 * E + dr * H); sp.ldx1 = L * E
            sp.gates = _ptr(gs[dr, t0]); sp.ldgs = 4 * H
            sp.c_prev = _ptr(cs[dr, t0 if dr == 0 else t0 + 1]); sp.ldcp = H
            sp.c_cur = _ptr(cs[dr, t0 + 1 if dr == 0 else t0]); sp.ldcc = H
            sp.dc = _ptr(dc_enc[dr]); sp.lddc = H
            sp.dg_next = _ptr(dgt[dr, S if dr == 0 else 0]); sp.lddg = 4 * H
            sp.dg_out = _ptr(dgt[dr, t0 if dr == 0 else t0 + 1]); sp.ldgo = 4 * H
            sp.dg_out2 = _ptr(dpre, t0 * 8 * H + dr * 4 * H); sp.ldgo2 = Lp * 8 * H
            sp.len = len32.data_ptr(); sp.t = t0
            ic = incs[dr]
            ic.dg = sg * B * 4 * H; ic.dg2 = sg * 8 * H; ic.ext1 = sg * E; ic.gates = sg * B * 4 * H
            ic.c_prev = sg * B * H; ic.c_cur = sg * B * H; ic.dt = sg
        # (S launches: a persistent, weight-stationary launch of this recurrence - the mirror of the forward's - was built in round 4
        #  and removed: 1.4 instead of 1.8 ms for the chain, but the step's tail is bound by the side stream's weight-gradient GEMMs,
        #  which then collide with the encoder convolutions' backward instead: 63.0 against 63.1 ms, profiles/r04_ab_bilstm_bwd_persistent.txt)
        call("t2_lstm_seq_bwd", steps, incs, 2, S, st)
        Rr = B * Lp - 4
        x3 = e["x3"]

        def bilstm_wgrads():
            for dr in range(2):
                nm = "encoder.lstm.weight_hh_l0" + ("" if dr == 0 else "_reverse")
                hprev = hs[0, 0] if dr == 0 else hs[1, 1]
                self._wgrad(_ptr(dgt[dr, 0 if dr == 0 else 1]), 4 * H, hprev, H, G[nm], H, 4 * H, H, S * B)
            self._wgrad(dpre, 8 * H, _ptr(x3, 2 * E), E, ps.cat_view("encoder.lstm.weight_ih_l0", 8 * H, E, grad=True), E, 8 * H, E, Rr)
            call("t2_colsum", dpre, 8 * H, B * Lp, 8 * H, ps.cat_view("encoder.lstm.bias_ih_l0", 8 * H, 0, grad=True), _stream())
            call("t2_colsum", dpre, 8 * H, B * Lp, 8 * H, ps.cat_view("encoder.lstm.bias_hh_l0", 8 * H, 0, grad=True), _stream())
        enc_wgrad_stream = side if self.defer_wgrads else None
        if enc_wgrad_stream is not None:     # the encoder's weight gradients leave the main stream too (it keeps the dgrad chain)
            ev = main.record_event()
            with torch.cuda.stream(side):
                side.wait_event(ev)
                bilstm_wgrads()
        else:
            bilstm_wgrads()
        dx = self.buf("enc.dx3", B * Lp, E)
        gemm_fill(dpre, ps.cat_view("encoder.lstm.weight_ih_l0", 8 * H, E), dx, Rr, E, 8 * H, 8 * H, E, E, a_k=1, b_k=0)

        self.mark("bwd.bilstm")
        # ---- encoder convolutions + embedding -----------------------------------------------------------------
        for li, i in reversed(list(enumerate((0, 4, 8)))):
            dx = self.conv_bn_bwd(f"enc.conv{li}", ctx, dx, Lp, 0, P[f"encoder.convolutions.{i}.weight"],
                                  G[f"encoder.convolutions.{i}.weight"], G[f"encoder.convolutions.{i}.bias"],
                                  f"encoder.convolutions.{i + 1}", B, L, E, E, 1, training, wgrad_stream=enc_wgrad_stream)
        call("t2_embedding_bwd", ctx["chars_idx"], dx, G["encoder.embedding.weight"], B, L, E, Lp, 0, st)
        torch.cuda.current_stream().wait_stream(self.side_stream())
        self.mark("bwd.encoder_convs")

    # =============================================================================================
    # loss + one optimisation step
    # =============================================================================================
    def loss_and_grads(self, outs, ctx, mel_tgt, gate_tgt, grad_scale=1.0):
        """3-term loss of model/tts_model.py:197-201 and the full backward.  Returns loss3 (device, float64[3])."""
        mels, post, gates, _ = outs
        B, T, M = mels.shape
        loss3 = self.buf("loss3", 3, dtype=torch.float64)
        d_post = self.buf("d_post", B, T, M)
        dproj = self.buf("dproj", T, B, M + 1)
        call("t2_loss_fwd_bwd", mels, post, gates, mel_tgt, gate_tgt, ctx["mlen32"], B, T, M, loss3, d_post, dproj,
             float(grad_scale), _stream())
        self.backward_tf(ctx, d_post, dproj)
        return loss3

    def adam_step(self, step, lr, weight_decay, max_norm=1.0, grad_scale=1.0, betas=(0.9, 0.999), eps=1e-8, ranges=None):
        """Global-norm clip + Adam(L2) on the flat buffers.  ranges: optional [(start, end)] element ranges to update (the
        trainable tensors when some are frozen; the caller zeroes the frozen gradients so the norm excludes them)."""
        ps = self.ps
        ps.init_adam()
        sumsq = self.buf("sumsq", 1, dtype=torch.float64)
        call("t2_sumsq", ps.grad, ps.numel, sumsq, _stream())
        for a, b in (ranges or [(0, ps.numel)]):
            call("t2_adam_step", _ptr(ps.flat, a), _ptr(ps.grad, a), _ptr(ps.exp_avg, a), _ptr(ps.exp_avg_sq, a), b - a, sumsq,
                 float(max_norm), float(lr), float(betas[0]), float(betas[1]), float(eps), float(weight_decay), int(step),
                 float(grad_scale), _stream())
        return sumsq

    # =============================================================================================
    # autoregressive inference: forward(teacher_forcing=False, max_len_override=N)  (model/tacotron2.py:262-325)
    # =============================================================================================
    def _infer_group(self, g, enc, chars_len, Tcap, speaker_id, description_embeddings, training, prenet_masks, controls):
        """Conditioning and decode-loop operands of one group of <= 64 utterances (workspaces prefixed inf<g>.); `enc` is the
        group's slice of the encoder output, computed for the WHOLE batch by the caller."""
        d, P, ps = self.d, self.ps.P, self.ps
        B, L = enc.shape[0], enc.shape[1]
        M, E, Pd, A, D, Ad = d["num_mels"], d["encoded_dim"], d["prenet_dim"], d["att_rnn_dim"], d["rnn_hidden_dim"], d["att_dim"]
        Ef = E + (128 if d.get("description_embeddings") else 0)
        F = d.get("loc_filters", 32)
        st = _stream()
        pf = f"inf{g}."
        len32 = chars_len.to(torch.int32)
        memory = self.buf(pf + "memory", B, L, Ef)
        desc = None
        if d.get("description_embeddings"):
            desc = self.buf(pf + "desc", B, 128)
            Dd = d["description_embeddings_dim"]
            gemm(description_embeddings, P["description_embeddings_linear.0.weight"], desc, B, 128, Dd, Dd, Dd, 128)
            call("t2_tanh_bias", desc, P["description_embeddings_linear.0.bias"], B, 128, st)
        spk32 = speaker_id.to(torch.int32) if d.get("speaker_tokens") else None
        call("t2_condition_fwd", enc, P["speaker_embedding.weight"] if d.get("speaker_tokens") else None, spk32, desc,
             memory, B, L, E, Ef, st)
        pmT = self.buf(pf + "pmT", B, Ad, L)
        gemm(P["att_encoder.weight"], memory, pmT, Ad, L, Ef, Ef, Ef, L, batch=B, sA=0, sB=L * Ef, sC=Ad * L)
        ldp = D + Ef
        ldo = (M + 1 + 3) // 4 * 4
        Bp = (B + 15) // 16 * 16
        xs = self.buf(pf + "xs", 2, (Pd + A + Ef + D) // 16, Bp, 16, zero=True)
        att_h = self.buf(pf + "att_h", B, A, zero=True)
        att_c = self.buf(pf + "att_c", 2, B, A, zero=True)
        dec_c = self.buf(pf + "dec_c", 2, B, D, zero=True)
        cum = self.buf(pf + "cum", 2, B, L, zero=True)
        xproj = self.buf(pf + "xproj", B, ldp)
        p1 = self.buf(pf + "p1", B, Pd)
        e_part = self.buf(pf + "e_part", B, Ad // 16, L)
        proj = self.buf(pf + "proj", Tcap, B, ldo)
        align = torch.zeros(B, Tcap, L, dtype=torch.float32, device=self.dev)
        done = self.buf(pf + "done", B, dtype=torch.int32, zero=True)
        state = self.buf(pf + "state", 2, dtype=torch.int32, zero=True)
        _, cterm, cmel1 = self.controls_terms(controls, B)
        row_comb = None
        if cmel1 is not None:        # per-utterance controls term of the folded linear: [W_pre1 . cmel_b ; cmel_b ; 0]
            row_comb = self.buf(pf + "row_comb", B, Pd + M + 1)
            gemm(cmel1, P["prenet.0.weight"], row_comb, B, Pd, M, M + 1, M, Pd + M + 1)
            row_comb[:, Pd:].copy_(cmel1)
            cterm = cterm.clone()    # (controls_terms reuses one workspace per engine)
        if prenet_masks is not None:
            pm = prenet_masks
        elif float(d["dropout"]) > 0.0:
            pm = self.buf(pf + "pmask", Tcap, 2, B, Pd)
        else:
            pm = None
        a = make("T2Infer", B=B, L=L, A=A, D=D, Ef=Ef, Ad=Ad, P=Pd, M=M, Kl=KL, Tcap=Tcap,
                 W_comb=self._w_comb, b_comb=self._b_comb, row_comb=row_comb, W_pre2=P["prenet.3.weight"],
                 W_comb_t=self._w_comb_t, W_pre2_t=self._w_pre2_t, p1_t=self.buf(pf + "p1_t", Pd // 16, Bp, 16, zero=True),
                 wp_att=self._wp_att_inf, b_att_ih=P["decoder.att_rnn.bias_ih"], b_att_hh=P["decoder.att_rnn.bias_hh"],
                 wp_dec=self._wp_dec_inf, b_dec_ih=P["decoder.lstm.bias_ih"], b_dec_hh=P["decoder.lstm.bias_hh"],
                 Wq=P["decoder.attention.query_layer.weight"], U=self._U_inf, v=P["decoder.attention.v.weight"],
                 pmT=pmT, memory=memory, len=len32, prenet_mask=pm, xs=xs, att_h=att_h, att_c=att_c, dec_c=dec_c, cum=cum,
                 xproj=xproj, p1=p1, e_part=e_part, proj=proj, ld_proj=ldo, align=align, done=done, state=state, dec_pre=cterm)
        return dict(a=a, B=B, proj=proj, align=align, state=state, pm=pm, philox=prenet_masks is None and pm is not None)

    def infer(self, chars_idx, chars_len, max_len, speaker_id=None, description_embeddings=None, training=False,
              prenet_masks=None, seed=0, check_every=32, controls=None):
        """Returns (mels, mels_post, gates, alignments, lengths) exactly as the reference's non-teacher-forced forward: ONE loop
        over the whole batch that ends when every utterance has produced a negative stop logit (model/tacotron2.py:319-322).
        Batches above 64 utterances are decoded as groups of 64 in lock-step chunks of `check_every` frames; the break frame
        and `lengths` are taken from the stored stop logits of all groups (t2_stop_scan).
        prenet_masks: optional [n][2][B][P] scale masks (parity tests); otherwise Philox masks (AlwaysDropout)."""
        d, P, ps = self.d, self.ps.P, self.ps
        B, L = chars_idx.shape
        assert B <= 4096, "engine.infer handles up to 4096 utterances per call (64 groups of 64)"
        self.generation += 1          # the encoder / postnet workspaces are shared with forward_tf
        self.begin_phase(backward=False)
        M, E, Pd, A, D = d["num_mels"], d["encoded_dim"], d["prenet_dim"], d["att_rnn_dim"], d["rnn_hidden_dim"]
        Ef = E + (128 if d.get("description_embeddings") else 0)
        F = d.get("loc_filters", 32)
        st = _stream()
        Tcap = int(max_len)
        if prenet_masks is not None:
            Tcap = min(Tcap, int(prenet_masks.shape[0]))     # only as many frames as masks were supplied
        ldp = D + Ef
        # ---- per-call operands shared by all groups ----
        self._U_inf = self.buf("U", d["att_dim"], 2, KL)
        call("t2_attn_fold_location", P["decoder.attention.location_dense.weight"],
             P["decoder.attention.location_conv.weight"], self._U_inf, d["att_dim"], F, KL, st)
        Wih_a, Wih_d = P["decoder.att_rnn.weight_ih"], P["decoder.lstm.weight_ih"]
        # packed in the column order of the tiled state [prenet | att_h | ctx | dec_h] (csrc/t2_infer.hip)
        self._wp_att_inf = self.pack_fwd("inf.att", [(Wih_a, Pd + Ef, Pd), (P["decoder.att_rnn.weight_hh"], A, A),
                                                     (_ptr(Wih_a, Pd), Pd + Ef, Ef)], A)
        self._wp_dec_inf = self.pack_fwd("inf.dec", [(Wih_d, A + Ef, A), (_ptr(Wih_d, A), A + Ef, Ef),
                                                     (P["decoder.lstm.weight_hh"], D, D)], D)
        # first prenet layer folded onto the mel projection: W_comb = [W_pre1 . W_mel ; W_mel ; W_gate] (include/tacotron2_amd.h)
        wproj = ps.cat_view("decoder.mel_out.weight", M + 1, ldp)
        bproj = ps.cat_view("decoder.mel_out.bias", M + 1, 0)
        self._w_comb = self.buf("inf.w_comb", Pd + M + 1, ldp)
        self._b_comb = self.buf("inf.b_comb", Pd + M + 1)
        gemm(P["prenet.0.weight"], wproj, self._w_comb, Pd, ldp, M, M, ldp, ldp, a_k=1, b_k=0)
        gemm(P["prenet.0.weight"], bproj, self._b_comb, Pd, 1, M, M, 1, 1, a_k=1, b_k=0)
        self._w_comb[Pd:].copy_(wproj); self._b_comb[Pd:].copy_(bproj)
        # x16-tiled copies of the two small linears' weights (data movement only): [K/16][rows padded to 16][16], the combined
        # linear's K chunks in the order [ctx | dec_h] of the tiled state
        Nc = Pd + M + 1
        Ncp = (Nc + 15) // 16 * 16
        wc = torch.zeros(Ncp, ldp, device=self.dev)
        wc[:Nc, :Ef] = self._w_comb[:, D:]; wc[:Nc, Ef:] = self._w_comb[:, :D]
        self._w_comb_t = wc.view(Ncp, ldp // 16, 16).permute(1, 0, 2).contiguous()
        self._w_pre2_t = P["prenet.3.weight"].view(Pd, Pd // 16, 16).permute(1, 0, 2).contiguous()
        # The encoder runs ONCE over the whole batch (model/tacotron2.py:197): in train() mode its BatchNorm layers then see the
        # statistics of all utterances, as in the reference, whatever the grouping of the decode loop below
        enc_all = self.encoder_fwd(chars_idx.contiguous(), chars_len.to(torch.int32), training, {}, {})
        groups = []
        for g, b0 in enumerate(range(0, B, 64)):
            sl = slice(b0, min(B, b0 + 64))
            groups.append(self._infer_group(
                g, enc_all[sl], chars_len[sl], Tcap,
                speaker_id[sl] if speaker_id is not None else None,
                description_embeddings[sl].contiguous() if description_embeddings is not None else None, training,
                prenet_masks[:, :, sl].contiguous() if (prenet_masks is not None and B > 64) else prenet_masks,
                controls[sl] if controls is not None else None))
        p = float(d["dropout"])
        t0 = 0
        self.mark("inf.encoder")
        # The host looks at the groups' "everyone has stopped" words once per chunk of `check_every` frames - and ONE CHUNK LATE:
        # the words of chunk k travel to pinned host memory behind the chunk (asynchronous copy + event) while chunk k+1 is already
        # enqueued, so the device never waits for the host's round trip (the reference synchronises `done.all()` every frame,
        # model/tacotron2.py:321).  A loop that has stopped therefore runs at most one chunk longer than it needed to; the exact
        # break frame and `lengths` come from the stored stop logits afterwards (t2_stop_scan), whatever was computed past it.
        pend = None
        pin = torch.empty(len(groups), 2, dtype=torch.int32, pin_memory=True)
        pins = [pin, pin.clone().pin_memory()]
        k = 0
        while t0 < Tcap:
            t1 = min(Tcap, t0 + check_every)
            for gi, G in enumerate(groups):
                if G["philox"]:
                    Bg = G["B"]
                    n = (t1 - t0) * 2 * Bg * Pd
                    call("t2_philox_mask", _ptr(G["pm"], t0 * 2 * Bg * Pd), n, p, seed, 1000 + t0 + 100003 * gi, st)
                call("t2_decoder_infer", G["a"], t0, t1, st)
            t0 = t1
            buf = pins[k & 1]
            for gi, G in enumerate(groups):
                buf[gi].copy_(G["state"], non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream())
            if pend is not None:
                pend[0].synchronize()
                if bool((pend[1][:, 0] != 0).all()):
                    break
            pend = (ev, buf)
            k += 1
        self.mark("inf.frame_loop")
        # exact break frame and lengths from the stored stop logits (model/tacotron2.py:319-322)
        lengths = self.buf("inf.lengths", B, dtype=torch.int64)
        nfr = self.buf("inf.nframes", 2, dtype=torch.int32)
        ldo = (M + 1 + 3) // 4 * 4
        scan = make("T2StopScan", proj=[G["proj"] for G in groups] + [0] * (64 - len(groups)),
                    Bg=[G["B"] for G in groups] + [0] * (64 - len(groups)), ngroups=len(groups), ld_proj=ldo, M=M, nframes=t0)
        call("t2_stop_scan", scan, lengths, nfr, st)
        n = max(int(nfr.cpu()[0]), 1)
        self.check_persistent_kernels()      # (the encoder recurrence is a persistent launch; the host has just synchronised anyway)
        # outputs: mask by the counted lengths, postnet on the unmasked mels (model/tacotron2.py:327-345)
        mlen32 = lengths.to(torch.int32)
        mels = torch.empty(B, n, M, dtype=torch.float32, device=self.dev)
        gates = torch.empty(B, n, 1, dtype=torch.float32, device=self.dev)
        post_in = self.buf("post.x0", B, n + 4, M)
        for G, b0 in zip(groups, range(0, B, 64)):
            Bg = G["B"]
            call("t2_finalize_fwd", G["proj"], ldo, mlen32[b0:b0 + Bg], mels[b0:b0 + Bg], gates[b0:b0 + Bg],
                 post_in[b0:b0 + Bg], Bg, n, M, st)
        Pn = d["postnet_dim"]
        chans = [M, Pn, Pn, Pn, Pn, M]
        x = post_in
        post = torch.empty(B, n, M, dtype=torch.float32, device=self.dev)
        pctx: dict = {}
        for li in range(5):
            last = li == 4
            x = self.conv_bn_fwd(f"post.conv{li}", x, P[f"postnet.postnet.{4 * li}.weight"], None,
                                 f"postnet.postnet.{4 * li + 1}", B, n, chans[li], chans[li + 1], 0 if last else 2, None,
                                 training, pctx, y=post if last else None, Lp_y=n if last else None,
                                 pad_y=0 if last else 2, res=post_in if last else None, Lp_res=n + 4, pad_res=2,
                                 length=mlen32 if last else None, fill=0.0)
        align = groups[0]["align"][:, :n] if len(groups) == 1 else torch.cat([G["align"][:, :n] for G in groups])
        return mels, post, gates, align.contiguous(), lengths.clone()
