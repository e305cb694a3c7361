"""Mirror of the reference's operator base class (skyplane/gateway/operators/gateway_operator.py:32-122) and the
GPU compress+hash operator that BASELINE.json's north_star asks for ("rewritten compress operator under
skyplane/gateway/operators").

GatewayOperator keeps the reference contract exactly:
  * ``start_workers`` forks ``n_processes`` ``Process(target=self.worker_loop, args=(i,)+self.args)`` (:66-70)
  * ``stop_workers`` sets the per-worker exit flags and joins (:72-77)
  * ``worker_loop``: get_nowait -> log in_progress -> ``process`` -> True: log complete + output_queue.put,
    False: re-queue after 0.1 s, exception: traceback to error_queue + error_event.set() (:79-115)
  * ``worker_exit`` hook (:117-118), abstract ``process`` (:120-122)

GatewayHipCompress replaces the two CPU call sites of the source gateway's hot path --
``lz4.frame.compress`` inside GatewaySender.process (:358-361) and the ``hashlib.md5`` loop requested by
GatewayObjStoreReadOperator.process (:555-565, result dropped at :577-582) -- with one batched call into
libskyhip.so.  It overrides ``worker_loop`` (the base loop sleeps 0.1 s per chunk and
busy-spins when idle, capping one worker below 80 MiB/s) to drain up to ``max_batch`` requests per iteration.
There is no CPU fallback: if the HIP extension or the GPU is missing the worker raises, which the loop turns
into the reference's error path (error_queue + error_event -> daemon-wide stop).
"""
from __future__ import annotations

import os
import queue
import random
import sys
import threading
import time
import traceback
from abc import ABC, abstractmethod
from multiprocessing import Event, Process, Queue
from typing import Callable, List, Optional

import numpy as np

from skyplane_amd.chunk import ChunkRequest, ChunkState
from skyplane_amd.gateway import dedup_wire, shm_arena, sidecar
from skyplane_amd.gateway.chunk_store import ChunkStore
from skyplane_amd.gateway.gateway_queue import GatewayQueue


_OP_TRACE = bool(os.environ.get("SKY_OP_TRACE"))


class GatewayOperator(ABC):
    yield_sleep_s = 0.1  # gateway_operator.py:102 "yield ?"

    def __init__(self, handle: str, region: str, input_queue: GatewayQueue, output_queue: GatewayQueue, error_event, error_queue: Queue,
                 chunk_store: ChunkStore, n_processes: Optional[int] = 1):
        self.handle = handle
        self.region = region
        self.input_queue = input_queue
        self.output_queue = output_queue
        self.chunk_store = chunk_store
        self.error_event = error_event
        self.error_queue = error_queue
        self.n_processes = n_processes
        self.args = ()
        self.processes: List[Process] = []
        self.exit_flags = [Event() for _ in range(self.n_processes)]
        self.worker_id: Optional[int] = None

    def start_workers(self):
        for i in range(self.n_processes):
            p = Process(target=self.worker_loop, args=(i,) + self.args)
            p.start()
            self.processes.append(p)

    def stop_workers(self):
        for i in range(self.n_processes):
            self.exit_flags[i].set()
        for p in self.processes:
            p.join()
        self.processes = []

    def _process_in_lane(self, batch, worker_id):
        """process_batch as the lane loop calls it.  On the dedup path (ONE lane per worker: the fingerprint table belongs to a context) the work that
        follows the device call -- recipe headers, payload and side-car files, completion records: ~100 ms of Python and file I/O per 64 chunks -- goes
        to a helper thread, and the lane reads and launches the next batch meanwhile (round 5; process_batch's `defer`)."""
        if not (self.dedup_wire and self.async_publish):
            return self.process_batch(batch)
        pool = getattr(self._tls, "publish_pool", None)
        if pool is None:
            from concurrent.futures import ThreadPoolExecutor

            pool = self._tls.publish_pool = ThreadPoolExecutor(1, thread_name_prefix=f"{self.handle}-publish")

        tries = self._tls.__dict__.setdefault("tries", {})      # (the LANE's back-off history: the helper thread has its own thread-local state)

        def defer(fn):
            def run():
                try:
                    for cr, meta in zip(batch, fn()):
                        self.chunk_store.log_chunk_state(cr, ChunkState.complete, operator_handle=self.handle, worker_id=worker_id, metadata=meta)
                        tries.pop(cr.chunk.chunk_id, None)      # as the synchronous path does: the dict must not grow with the transfer
                        if self.output_queue is not None:
                            self.output_queue.put(cr)
                except Exception:
                    self.error_queue.put(traceback.format_exc())
                    self.error_event.set()
                    raise
            return pool.submit(run)

        return self.process_batch(batch, defer=defer)

    def worker_loop(self, worker_id: int, *args):
        self.worker_id = worker_id
        while not self.exit_flags[worker_id].is_set() and not self.error_event.is_set():
            try:
                try:
                    chunk_req = self.input_queue.get_nowait(self.handle)
                except queue.Empty:
                    continue
                self.chunk_store.log_chunk_state(chunk_req, ChunkState.in_progress, operator_handle=self.handle, worker_id=worker_id)
                succ = self.process(chunk_req, *args)
                if succ:
                    self.chunk_store.log_chunk_state(chunk_req, ChunkState.complete, operator_handle=self.handle, worker_id=worker_id)
                    if self.output_queue is not None:
                        self.output_queue.put(chunk_req)
                    time.sleep(self.yield_sleep_s)
                else:
                    time.sleep(0.1)
                    self.input_queue.put(chunk_req)
            except Exception:
                self.error_queue.put(traceback.format_exc())
                self.error_event.set()
                self.exit_flags[worker_id].set()
        self.worker_exit(worker_id)

    def worker_exit(self, worker_id: int):
        pass

    @abstractmethod
    def process(self, chunk_req: ChunkRequest, **args):
        pass


def _default_context_factory(device_id: int, max_chunk_bytes: int, max_batch: int):
    from skyplane_amd import hip_ops  # imported in the forked worker only: HIP must not be initialised in the daemon parent

    return hip_ops.SkyHipContext(device_id=device_id, max_chunk_bytes=max_chunk_bytes, max_batch=max_batch)


def visible_gpu_count() -> int:
    """Number of GPUs without touching the HIP runtime in this (possibly parent) process."""
    env = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
    if env:
        return len([x for x in env.split(",") if x.strip() != ""])
    try:
        return max(1, len([d for d in os.listdir("/sys/class/kfd/kfd/topology/nodes")
                           if open(f"/sys/class/kfd/kfd/topology/nodes/{d}/gpu_id").read().strip() not in ("", "0")]))
    except OSError:
        return 1


class GatewayHipCompress(GatewayOperator):
    """op_type "gpu_compress": sits on the edge read_object_store -> [mux] -> send of the source gateway.

    For every ChunkRequest the raw bytes are at ``chunk_store.get_chunk_file_path(chunk_id)``.  The operator
    leaves that file untouched (GatewaySender asserts its size, :352) and writes next to it
      <id>.chunk.lz4f  -- a conformant LZ4 frame of the chunk (what lz4.frame.compress would have produced, up
                          to compressed bytes; lz4.frame.decompress at gateway_receiver.py:196 returns the raw bytes)
      <id>.chunk.md5   -- hex MD5 of the raw bytes (Chunk.md5_hash cannot carry bytes through the sender's
                          json.dumps at :299)
    and reports ``compressed_size_bytes`` / ``uncompressed_size_bytes`` in the status metadata consumed by the
    reference's compression profile endpoint (gateway_daemon_api.py:129-134, :340-354).
    ``max_batch`` should be large: whole-chunk MD5 is a serial chain (about 0.1 s per 8 MiB on one GPU lane whatever the
    batch size), so a worker moves at most max_batch x chunk / chain-time; LZ4 for the same batch takes milliseconds.
    Worker i binds GPU ``device_ids[i % len(device_ids)]``: chunks are independent, so the N GPUs of a node are N
    workers pulling from one queue -- no collective anywhere.
    """

    def __init__(self, handle: str, region: str, input_queue: GatewayQueue, output_queue: GatewayQueue, error_event, error_queue: Queue,
                 chunk_store: ChunkStore, n_processes: Optional[int] = 1, max_batch: int = 64, max_chunk_bytes: int = 64 << 20,
                 device_ids: Optional[List[int]] = None, compute_md5: bool = True, cdc: bool = False, dedup: bool = False,
                 idle_sleep_s: float = 0.001, context_factory: Optional[Callable] = None, pipeline_depth: Optional[int] = None, fill_wait_s: Optional[float] = None,
                 prealloc: bool = False, dedup_wire: bool = False, dedup_epoch_bytes: int = 1 << 30, handoff: str = "arena", arena_slots: int = 0,
                 arena_slot_bytes: int = 0, in_slots: int = 0, in_slot_chunk_bytes: int = 0):
        super().__init__(handle, region, input_queue, output_queue, error_event, error_queue, chunk_store, n_processes)
        # The RAW side of the source (round 6, SURVEY 8f item 2 "and the reader's file write"): `in_slots` page-locked slot files per worker process
        # (shm_arena.InSlots) that the reader -- GatewayObjStoreReadOperator with INTEGRATION 6e's two lines -- links ``<id>.chunk`` to BEFORE it downloads,
        # so that the chunk's bytes land in pinned memory and this operator hands them to the device where they lie.  0 = off (the default: it needs the
        # reader-side patch; a chunk file that is not a slot link is read as before, so turning it on without the patch costs only the pinned memory).
        # `in_slot_chunk_bytes`: the transfer's chunk size when it is known up front (then the slots exist before the first download), 0 = the first
        # batch's usual chunk length.  At most `in_slot_bytes` of slots per worker.
        self.in_slots = int(in_slots)
        self.in_slot_chunk_bytes = int(in_slot_chunk_bytes)
        self.in_slot_bytes = 4 << 30
        self._in_slot_set = None               # (per worker process, shared by its lanes)
        self._in_slot_failed = False
        self._in_slots_lock = threading.Lock()
        # How a frame reaches the sender.  "arena": the device writes it by DMA into a slot of a shared, page-locked arena file in
        # the chunk directory and `<id>.chunk.lz4f` is a pointer to the slot (gateway/shm_arena.py) -- the sender sendfile()s it from there and unlinks
        # the pointer, which frees the slot.  "files": one payload file per chunk, as in rounds 1-2 (also what a chunk falls back to when every slot
        # is still waiting for its sender).  `arena_slots` per lane (0 = 2 x max_batch).
        assert handoff in ("arena", "files")
        self.handoff = handoff
        self.arena_slots = int(arena_slots)
        # bytes per arena slot: 0 = the frame bound of the largest chunk of the lane's FIRST batch (an arena is page-locked as a whole: sized for the
        # 64 MiB default chunk it was 4 GiB of pinned tmpfs per lane even when the transfer's chunks are 8 MiB -- ADVICE r3); a later, larger chunk goes
        # out as a plain payload file
        self.arena_slot_bytes = int(arena_slot_bytes)
        # dedup on the wire (skyplane_amd/gateway/dedup_wire.py): every chunk leaves as a recipe -- literal segments, LZ4-compressed, plus references to
        # segments this lane sent before -- instead of as one LZ4 frame.  Needs gpu_decompress(dedup_wire=True) on the destination gateway.
        self.dedup_wire = bool(dedup_wire)
        self.dedup_epoch_bytes = int(dedup_epoch_bytes)      # a lane empties its device table (and starts a new epoch) after this many chunk bytes
        if self.dedup_wire:
            cdc = dedup = True
        self.max_batch = int(max_batch)
        self.max_chunk_bytes = int(max_chunk_bytes)
        self.device_ids = list(device_ids) if device_ids is not None else list(range(visible_gpu_count()))
        self.compute_md5 = compute_md5
        self.cdc = cdc
        self.dedup = dedup
        self.idle_sleep_s = idle_sleep_s
        self._context_factory = context_factory or _default_context_factory
        # Every worker process runs `pipeline_depth` lanes (threads), each with its own device context and pinned arenas.  A batch's
        # whole-chunk MD5 is a serial chain of ~0.1 s per 8 MiB whatever the batch size: with one lane the worker would sit
        # in that call while the next batch waits; with several, lane B uploads and compresses while lane A's chain runs (the C ABI is
        # synchronous per context and ctypes releases the GIL, so lanes overlap on the device and in the kernel's file I/O).
        # Default 2 (round 5; 3 before): the loopback moves 33.6 / 44.0 / 38.1 Gbit/s with 1 / 2 / 3 lanes per worker (profiles/r5_operator_lanes.txt) -- every
        # lane's context brings six HIP streams, a process's streams share GPU_MAX_HW_QUEUES (4) hardware queues, and kernels of streams that share a
        # queue serialise: a third lane mostly queues behind the other two's 80 ms digest launches.  (More hardware queues are no way out: at 24 the same
        # loopback fell to 17.7 Gbit/s, although the bare device calls of scripts/host_path_bench.py then run three lanes at one lane's rate.)
        # ONE lane when the source deduplicates on the wire: every lane owns a device context with its own fingerprint table (up to GiBs of
        # HBM each) and duplicates that land on different lanes are never matched, so several lanes cost memory and hit rate.
        self.pipeline_depth = max(1, int(pipeline_depth)) if pipeline_depth is not None else (1 if self.dedup_wire else 2)
        # a lane that finds fewer than max_batch requests keeps collecting for up to this long before it launches: trickling input otherwise turns into
        # many small calls that each pay the full chain latency (~80 ms).  30 ms with three lanes per worker (another lane is on the device meanwhile);
        # 4 ms for the single lane of the dedup path, where a waiting lane is an idle device (profiles/r3_e2e_fill_wait.txt)
        self.fill_wait_s = float(fill_wait_s) if fill_wait_s is not None else (0.004 if self.dedup_wire else 0.03)
        # size every lane's pinned arenas for max_batch chunks of max_chunk_bytes when the lane starts, instead of growing them under the first full
        # batches: pinning fresh host memory runs at a few GB/s, which a transfer of seconds would otherwise pay inside its first batches
        self.prealloc = bool(prealloc)
        self.async_publish = True              # dedup path: a batch is published by a helper thread while the lane launches the next one (_process_in_lane)
        self.read_threads = 4                  # threads that copy a batch's chunk files into pinned staging (1 = one after the other, rounds 1-4)
        self._tls = threading.local()

    # -- process-local ---------------------------------------------------------------------------------
    @property
    def _ctx(self):
        return getattr(self._tls, "ctx", None)

    @_ctx.setter
    def _ctx(self, v):
        self._tls.ctx = v

    @property
    def _arenas(self):
        if not hasattr(self._tls, "arenas"):
            self._tls.arenas = {}
        return self._tls.arenas

    @_arenas.setter
    def _arenas(self, v):
        self._tls.arenas = v

    @property
    def _last_metadata(self):
        return getattr(self._tls, "last_metadata", [])

    @_last_metadata.setter
    def _last_metadata(self, v):
        self._tls.last_metadata = v

    def _context(self):
        if self._ctx is None:
            wid = self.worker_id or 0
            dev = self.device_ids[wid % len(self.device_ids)]
            self._ctx = self._context_factory(dev, self.max_chunk_bytes, self.max_batch)
        return self._ctx

    def _flags(self) -> int:
        return 1 | (2 if self.compute_md5 else 0) | (4 if self.cdc else 0) | (8 if self.dedup and self.cdc else 0)

    def _arena(self, ctx, which: str, nbytes: int):
        """Grow-only pinned staging area of this worker (skyhip_host_alloc): chunk files are read straight into it
        and the frames come back into it, so the only host copies left are the page-cache read and write."""
        cur = self._arenas.get(which)
        if cur is None or cur.size < nbytes:
            if cur is not None:
                ctx.release_pinned(cur)
            cur = ctx.pinned_buffer(max(nbytes, 1))
            self._arenas[which] = cur
        return cur

    def _writer(self, ctx, largest: int = 0) -> "shm_arena.ArenaWriter":
        """This lane's arena (created on first use, page-locked through the lane's context when that is a real device).  `largest` = the largest chunk of
        the batch at hand: it sizes the slots when arena_slot_bytes was left at 0.  A lane whose FIRST batch held only small chunks (small objects, a
        file's short tail) is not stuck with small slots for the rest of the transfer: when a larger chunk shows up the lane opens a new arena
        sized for it -- and says so -- while the old one stays mapped until its published slots have been sent."""
        fb = ctx.frame_bound if hasattr(ctx, "frame_bound") else (lambda n: 15 + n + 4 * ((n + 65535) // 65536) + 4)
        w = getattr(self._tls, "writer", None)
        # an arena this lane has outgrown goes -- mapping, page-lock and file -- as soon as the sender has sent (and deleted the pointer of) its last slot,
        # not at worker exit
        still = []
        for ow in getattr(self._tls, "old_writers", []):
            if any(o is not None and o.exists() for o in ow._owner):
                still.append(ow)
            else:
                shm_arena.forget(ow.arena)
                ow.arena.close(unlink=True)
        if hasattr(self._tls, "old_writers"):
            self._tls.old_writers = still
        need = (fb(min(max(largest, 1 << 16), self.max_chunk_bytes)) + 4095) & ~4095
        if w is not None and (self.arena_slot_bytes or need <= w.arena.slot_bytes):
            return w
        gen = getattr(self._tls, "writer_gen", 0)
        if w is not None:
            print(f"[{self.handle}] arena slots of {w.arena.slot_bytes} bytes (sized by this lane's first batch) are too small for a chunk of {largest} bytes: "
                  f"new arena with slots of {need} bytes", flush=True)
            old = getattr(self._tls, "old_writers", [])
            old.append(w)
            self._tls.old_writers = old
        bound = self.arena_slot_bytes or need
        tag = f"{self.handle}_{os.getpid()}_{threading.get_ident() & 0xFFFFFF:x}" + (f"_g{gen}" if gen else "")
        w = shm_arena.ArenaWriter(self.chunk_store.get_chunk_file_path("x").parent, tag, bound, self.arena_slots or 2 * self.max_batch)
        shm_arena.register_once(w.arena, ctx)
        self._tls.writer, self._tls.writer_gen = w, gen + 1
        return w

    def _in_slots(self, ctx, sizes) -> Optional["shm_arena.InSlots"]:
        """This WORKER's source slot files (made by its first lane that gets here; page-locked once per process, which is what every lane's context of the
        process may DMA from).  Sized by `in_slot_chunk_bytes` -- the transfer's chunk size, which the planner knows -- or, when that was left
        at 0, by the first batch's usual chunk length: the reader can only use them for the chunks it downloads AFTER they exist."""
        if self.in_slots <= 0:
            return None
        with self._in_slots_lock:
            if self._in_slot_set is None and not self._in_slot_failed:
                lens_sorted = sorted(s for s in sizes if s > 0)
                size = self.in_slot_chunk_bytes or (max(set(lens_sorted), key=lens_sorted.count) if lens_sorted else 0)
                if size <= 0:
                    return None
                n = max(2, min(self.in_slots, self.in_slot_bytes // size))
                ins = None
                try:
                    ins = shm_arena.InSlots(self.chunk_store.get_chunk_file_path("x").parent, f"{self.handle}_{os.getpid()}", size, n)
                    ins.register(ctx)
                    self._in_slot_set = ins
                except (OSError, MemoryError, RuntimeError) as e:
                    if ins is not None:
                        ins.close()
                    self._in_slot_failed = True
                    print(f"[{self.handle}] no page-locked source slots ({n} x {size} bytes: {type(e).__name__}: {e}): chunk files are read as before", flush=True)
            return self._in_slot_set

    def _read_chunks(self, chunk_reqs: List[ChunkRequest], ctx):
        """Raw bytes of every request.  With a real context: views of the pinned arena filled by readinto (zero-copy
        hand-off); otherwise plain bytes, as the reference reads them at gateway_operator.py:350-351."""
        sizes = [int(cr.chunk.chunk_length_bytes) for cr in chunk_reqs]
        pinned = hasattr(ctx, "pinned_buffer")
        arena, pos, datas = None, 0, []
        # Round 6: a chunk file that is a hard link to one of this worker's page-locked source slots (shm_arena.InSlots: the reader wrote the chunk INTO
        # pinned memory) is handed to the device where it lies -- no read, no copy.  Everything else is read into pinned staging as before.
        ins = self._in_slots(ctx, sizes)
        slot_views = [None] * len(chunk_reqs)
        if ins is not None:
            for j, (cr, size) in enumerate(zip(chunk_reqs, sizes)):
                try:
                    st = os.stat(self.chunk_store.get_chunk_file_path(cr.chunk.chunk_id))
                except FileNotFoundError:
                    continue
                if st.st_nlink >= 2:
                    slot_views[j] = ins.view_of(st, size)
        if pinned:
            arena = self._arena(ctx, "in", sum((s + 255) & ~255 for s, v in zip(sizes, slot_views) if v is None) or 1)
        jobs = []
        for cr, size, sv in zip(chunk_reqs, sizes, slot_views):
            if sv is not None:
                jobs.append((cr, size, sv, True))
                continue
            jobs.append((cr, size, arena[pos:pos + size] if pinned else None, False))
            pos += (size + 255) & ~255
        self._tls.__dict__["in_slot_hits"] = self._tls.__dict__.get("in_slot_hits", 0) + sum(v is not None for v in slot_views)

        def read_one(job):
            cr, size, data, in_place = job
            if in_place:
                return data[:size]
            path = self.chunk_store.get_chunk_file_path(cr.chunk.chunk_id)
            with open(path, "rb") as f:
                if data is None:
                    data = f.read()
                    got = len(data)
                else:
                    got, view = 0, memoryview(data)
                    while got < size:
                        k = f.readinto(view[got:])
                        if not k:
                            break
                        got += k
                    got += len(f.read(1))     # a longer file is as wrong as a shorter one
            # same invariant GatewaySender.process asserts at gateway_operator.py:352
            assert got == size, f"chunk {cr.chunk.chunk_id} has size {got}{'+' if got > size else ''} but should be {size}"
            return data

        # the page cache -> pinned staging copies of a batch run side by side (readinto releases the GIL): a lane that is alone on its worker -- the
        # dedup path -- otherwise spends as long reading 64 chunk files one after the other (~100 ms) as the device spends on them
        if pinned and len(jobs) > 1 and self.read_threads > 1:
            pool = getattr(self._tls, "read_pool", None)
            if pool is None:
                from concurrent.futures import ThreadPoolExecutor

                pool = self._tls.read_pool = ThreadPoolExecutor(self.read_threads, thread_name_prefix=f"{self.handle}-read")
            return list(pool.map(read_one, jobs))
        return [read_one(j) for j in jobs]

    def process_batch(self, chunk_reqs: List[ChunkRequest], defer=None) -> Optional[List[bool]]:
        """defer (the lane loop's, dedup path on a real device only): a callable that takes the batch's finishing work -- recipe encoding, payload and
        side-car files, completion records -- to a helper thread; the call then returns None as soon as the device is done with the batch, and the lane
        starts the next one meanwhile.  The staging areas the helper still reads alternate between two sets (it is at most one batch behind)."""
        ctx = self._context()
        deferred = defer is not None and self.dedup_wire and hasattr(ctx, "dedup_literals") and hasattr(ctx, "pinned_buffer")
        par = ""
        if deferred:
            par = str(getattr(self._tls, "parity", 0))
            self._tls.parity = 1 - int(par)
            prev = self._tls.__dict__.setdefault("finishing", {}).pop(par, None)
            if prev is not None:
                prev.result()                          # the batch before last has left this set of staging areas
        datas = self._read_chunks(chunk_reqs, ctx)
        # frames that go out as they are (no recipes) are produced straight into arena slots when there are free ones
        writer = self._writer(ctx, max((len(d) for d in datas), default=0)) if (self.handoff == "arena" and not self.dedup_wire) else None
        fits = writer is not None and all((ctx.frame_bound(len(d)) if hasattr(ctx, "frame_bound") else len(d) + len(d) // 255 + 64) <= writer.arena.slot_bytes for d in datas)
        slots = writer.take(len(datas)) if fits else []
        published = 0
        try:
            if hasattr(ctx, "pinned_buffer") and hasattr(ctx, "frame_bound"):
                bounds = [ctx.frame_bound(len(d)) for d in datas]
                rest = bounds[len(slots):]
                out = self._arena(ctx, "out" + par, sum((b + 255) & ~255 for b in rest)) if rest else None
                views, pos = [writer.arena.slot(s)[:b] for s, b in zip(slots, bounds)], 0
                for b in rest:
                    views.append(out[pos:pos + b])
                    pos += (b + 255) & ~255
                results = ctx.process_batch(datas, flags=self._flags(), frames_into=views)
            else:
                results = ctx.process_batch(datas, flags=self._flags())
                for s, res in zip(slots, results):                # a context without DMA targets (emulator): one copy into the slot
                    writer.arena.slot(s)[:len(res.frame)] = np.frombuffer(res.frame, np.uint8)
        except BaseException:
            for s in slots:
                writer.give_back(s)
            raise
        recipes = self._build_recipes(ctx, datas, results, lit_name="lit" + par) if self.dedup_wire else None
        if deferred:
            sizes = [len(d) for d in datas]
            self._tls.finishing[par] = defer(lambda: self._publish(chunk_reqs, sizes, results, recipes, writer, slots))
            return None
        self._last_metadata = self._publish(chunk_reqs, [len(d) for d in datas], results, recipes, writer, slots)
        return [True] * len(chunk_reqs)

    def _publish(self, chunk_reqs, sizes, results, recipes, writer, slots) -> List[dict]:
        """Payload (pointer file, or file) + side-cars of every chunk of a batch the device is done with; returns the status metadata per chunk."""
        metas, published = [], 0
        try:
            for k, (cr, size, res) in enumerate(zip(chunk_reqs, sizes, results)):
                cid = cr.chunk.chunk_id
                payload = recipes[k][0] if recipes is not None else res.frame
                if k < len(slots):
                    writer.publish(slots[k], sidecar.compressed_path(self.chunk_store, cid), len(payload))      # pointer file: complete when visible
                    published = k + 1
                else:
                    tmp = sidecar.compressed_path(self.chunk_store, cid).with_suffix(".tmp")
                    with open(tmp, "wb") as f:
                        if hasattr(payload, "write_to"):
                            payload.write_to(f)           # a recipe in two pieces: its literal frame goes out from where the device put it
                        else:
                            f.write(payload)
                    os.replace(tmp, sidecar.compressed_path(self.chunk_store, cid))   # the sender never sees a partial frame
                meta = {"compressed_size_bytes": len(payload), "uncompressed_size_bytes": size}
                if recipes is not None:
                    meta["dedup_reference_bytes"] = recipes[k][1]
                if res.md5 is not None:
                    sidecar.digest_path(self.chunk_store, cid).write_text(res.md5.hex())
                    meta["md5_hex"] = res.md5.hex()
                if res.cuts is not None:
                    meta["cdc_segments"] = int(len(res.cuts))
                metas.append(meta)
        except BaseException:
            for sl in slots[published:]:          # slots taken for this batch and never published would stay busy for ever
                writer.give_back(sl)
            raise
        return metas

    def _build_recipes(self, ctx, datas, results, lit_name: str = "lit"):
        """One (payload, referenced bytes) per chunk of the device call that just returned (its CDC results are still the context's last ones)."""
        st = self._tls.__dict__.setdefault("dedup_state", {"lane": random.getrandbits(64), "epoch": 0, "bytes": 0})
        in_len = np.array([len(d) for d in datas], np.uint64)
        prefix, cuts, fps, first, base = ctx.cdc_results(len(datas), in_len)
        plans, lit_bufs, lit_owner = [], [], []
        if hasattr(ctx, "dedup_literals") and hasattr(ctx, "pinned_buffer"):
            # round 5: the literal streams are put together and compressed ON THE DEVICE, from the chunks the call above left resident there -- no gather
            # on the host, no second upload (rounds 2-4: np.concatenate of every chunk's new segments into pinned staging + a second, LZ4-only call)
            bounds = [ctx.frame_bound(len(d)) for d in datas]
            lit_arena = self._arena(ctx, lit_name, sum((b + 255) & ~255 for b in bounds))
            views, pos = [], 0
            for b in bounds:
                views.append(lit_arena[pos:pos + b])
                pos += (b + 255) & ~255
            t_lit = time.perf_counter()
            lit_lens, lit_frames = ctx.dedup_literals([len(d) for d in datas], views)
            if _OP_TRACE:
                print(f"[op-trace] {self.handle}: literal streams of {len(datas)} chunks on the device in {1e3 * (time.perf_counter() - t_lit):.1f} ms", file=sys.stderr, flush=True)
            out = []
            for i, (data, res) in enumerate(zip(datas, results)):
                lens, kinds, sl = dedup_wire.classify_segments(prefix, cuts, first, base, i)
                nlit = int(lens[kinds == dedup_wire.KIND_LITERAL].sum())
                assert nlit == lit_lens[i], f"chunk {i}: device literal stream of {lit_lens[i]} bytes, the segment list says {nlit}"
                frame = res.frame if nlit == len(data) else (lit_frames[i] if lit_frames[i] is not None else b"")
                out.append((dedup_wire.encode_recipe_parts(st["lane"], st["epoch"], lens, kinds, fps[sl], frame, nlit), int(len(data) - nlit)))
            st["bytes"] += int(in_len.sum())
            if st["bytes"] >= self.dedup_epoch_bytes:
                ctx.dedup_reset()
                st["epoch"] += 1
                st["bytes"] = 0
            return out
        # (contexts without the device gather -- emulator, test doubles --) the literal streams of chunks with duplicates are gathered straight into
        # pinned staging (the second device call then uploads asynchronously)
        lit_arena = self._arena(ctx, "lit", sum((len(d) + 255) & ~255 for d in datas)) if hasattr(ctx, "pinned_buffer") else None
        lit_pos = 0
        for i, (data, res) in enumerate(zip(datas, results)):
            lens, kinds, sl = dedup_wire.classify_segments(prefix, cuts, first, base, i)
            if not kinds.any():                        # nothing to leave out: the frame the compressor made IS the literal stream
                plans.append([lens, kinds, fps[sl], res.frame, len(data)])
                continue
            arr = data if isinstance(data, np.ndarray) else np.frombuffer(data, np.uint8)
            ends = np.cumsum(lens.astype(np.int64))
            parts = [arr[e - l:e] for e, l, kd in zip(ends, lens, kinds) if kd == dedup_wire.KIND_LITERAL]
            if parts and lit_arena is not None:
                nlit = int(sum(p.size for p in parts))
                lit = lit_arena[lit_pos:lit_pos + nlit]
                np.concatenate(parts, out=lit)
                lit_pos += (nlit + 255) & ~255
            else:
                lit = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
            plans.append([lens, kinds, fps[sl], b"", int(lit.size)])
            if lit.size:
                lit_bufs.append(lit)
                lit_owner.append(i)
        if lit_bufs:      # second device call, LZ4 only: the fingerprint table is not touched
            for i, r in zip(lit_owner, ctx.process_batch(lit_bufs, flags=1)):
                plans[i][3] = r.frame
        out = [(dedup_wire.encode_recipe(st["lane"], st["epoch"], lens, kinds, fp, frame, nlit), int(lens[kinds == dedup_wire.KIND_REFERENCE].sum()))
               for lens, kinds, fp, frame, nlit in plans]
        st["bytes"] += int(in_len.sum())
        if st["bytes"] >= self.dedup_epoch_bytes:      # bound what the destination has to remember: new epoch, empty table
            ctx.dedup_reset()
            st["epoch"] += 1
            st["bytes"] = 0
        return out

    def process(self, chunk_req: ChunkRequest, **args):
        return self.process_batch([chunk_req])[0]

    def _parked(self) -> list:
        """This lane's not-ready requests: [due time, attempts, request].  They wait HERE with an exponential back-off (10 ms ... 0.5 s) instead of
        going straight back into the shared queue, where the next loop iteration would pick them up again at once."""
        if not hasattr(self._tls, "parked"):
            self._tls.parked = []
        return self._tls.parked

    def _park(self, cr: ChunkRequest):
        parked = self._parked()
        tries = getattr(self._tls, "tries", None)
        if tries is None:
            tries = self._tls.tries = {}
        n = tries.get(cr.chunk.chunk_id, 0)
        tries[cr.chunk.chunk_id] = n + 1
        parked.append([time.monotonic() + min(0.5, 0.01 * (1 << min(n, 6))), n, cr])

    def _take_batch(self) -> List[ChunkRequest]:
        batch: List[ChunkRequest] = []
        parked = self._parked()
        if parked:
            now = time.monotonic()
            due = [e for e in parked if e[0] <= now]
            for e in due[: self.max_batch]:
                parked.remove(e)
                batch.append(e[2])
        # A device call costs ~80 ms whatever it holds (the MD5 of an 8 MiB chunk is one serial chain), so a call per straggler is the expensive way to be
        # prompt: once there is something to do, keep collecting for up to fill_wait_s (profiles/r3_e2e_fill_wait.txt), less when the batch fills up
        deadline = None
        while len(batch) < self.max_batch:
            try:
                batch.append(self.input_queue.get_nowait(self.handle))
            except queue.Empty:
                if not batch or self.fill_wait_s <= 0:
                    break
                now = time.monotonic()
                if deadline is None:
                    deadline = now + self.fill_wait_s
                if now >= deadline:
                    break
                time.sleep(min(0.002, deadline - now))
        return batch

    def _prealloc(self):
        ctx = self._context()
        if not (hasattr(ctx, "pinned_buffer") and hasattr(ctx, "frame_bound")):
            return
        per = (ctx.frame_bound(self.max_chunk_bytes) + 255) & ~255
        # with the shared-memory hand-off the payload side needs no private staging: the compressor's frames go to its arena (made and page-locked
        # here), the decompressor's payloads come from the receiver's
        decomp = isinstance(self, GatewayHipDecompress)
        arena_side = "in" if decomp else "out"
        names = ["in", "out"]
        if self.dedup_wire and not decomp and getattr(self, "async_publish", False) and hasattr(ctx, "dedup_literals"):
            # the deferred-publish path stages in two alternating sets (process_batch: "out" + parity, "lit" + parity): those are what the first batches
            # would otherwise pin inside the timed path, and a plain "out" would stay pinned unused
            names = ["in", "out0", "out1", "lit0", "lit1"]
        for which in names:
            if self.handoff == "arena" and not self.dedup_wire and which == arena_side:
                continue
            self._arena(ctx, which, per * self.max_batch)[::4096] = 0
        if self.handoff == "arena" and not self.dedup_wire and not decomp:
            self._writer(ctx, self.max_chunk_bytes)      # prealloc: the configured maximum sizes the slots


    def _lane_loop(self, worker_id: int):
        """One pipeline lane: drain up to max_batch requests, one device call, hand the chunks on."""
        if self.prealloc or (self.in_slots > 0 and self.in_slot_chunk_bytes > 0):
            try:
                if not isinstance(self, GatewayHipDecompress) and self.in_slots > 0 and self.in_slot_chunk_bytes > 0:
                    self._in_slots(self._context(), [self.in_slot_chunk_bytes])      # the reader can claim a slot for the very first chunk
                if self.prealloc:
                    self._prealloc()
            except Exception:
                self.error_queue.put(traceback.format_exc())
                self.error_event.set()
                self.exit_flags[worker_id].set()
        while not self.exit_flags[worker_id].is_set() and not self.error_event.is_set():
            try:
                batch = self._take_batch()
                if not batch:
                    time.sleep(self.idle_sleep_s)      # no busy spin (reference :84-88 spins)
                    continue
                for cr in batch:
                    self.chunk_store.log_chunk_state(cr, ChunkState.in_progress, operator_handle=self.handle, worker_id=worker_id)
                oks = self._process_in_lane(batch, worker_id)
                if oks is None:
                    continue                           # the batch is being published by the lane's helper thread, which also records its completion
                retry = []
                for cr, ok, meta in zip(batch, oks, self._last_metadata):
                    if ok:
                        self.chunk_store.log_chunk_state(cr, ChunkState.complete, operator_handle=self.handle, worker_id=worker_id, metadata=meta)
                        getattr(self._tls, "tries", {}).pop(cr.chunk.chunk_id, None)      # (its back-off history goes with it: the dict must not grow with the transfer)
                        if self.output_queue is not None:
                            self.output_queue.put(cr)
                    else:
                        retry.append(cr)
                for cr in retry:
                    self._park(cr)                     # comes back to this lane after its back-off (the reference sleeps 0.1 s and re-queues, :103-106)
                if retry and len(retry) == len(batch):
                    time.sleep(self.idle_sleep_s)      # nothing could be done this round
            except Exception:
                self.error_queue.put(traceback.format_exc())
                self.error_event.set()
                self.exit_flags[worker_id].set()
        for fut in list(getattr(self._tls, "finishing", {}).values()):      # batches still being published by the helper thread
            try:
                fut.result()
            except Exception:
                pass                                   # (reported by the helper itself)
        for _due, _n, cr in self._parked():           # what still waits goes back to the queue: another worker (or a restart) may finish it
            self.input_queue.put(cr)
        self._parked().clear()
        getattr(self._tls, "tries", {}).clear()
        self.worker_exit(worker_id)                    # this lane's context and arenas (thread-local)

    def worker_loop(self, worker_id: int, *args):
        self.worker_id = worker_id
        lanes = [threading.Thread(target=self._lane_loop, args=(worker_id,), name=f"{self.handle}-w{worker_id}-lane{k}", daemon=True)
                 for k in range(self.pipeline_depth - 1)]
        for t in lanes:
            t.start()
        self._lane_loop(worker_id)                     # the worker's own thread is lane 0
        for t in lanes:
            t.join()
        self.process_exit(worker_id)

    def process_exit(self, worker_id: int):
        """Every lane of this worker process has left its loop."""
        ins, self._in_slot_set = self._in_slot_set, None
        if ins is not None:
            ins.close()

    def worker_exit(self, worker_id: int):
        for w in [getattr(self._tls, "writer", None)] + list(getattr(self._tls, "old_writers", [])):
            if w is None:
                continue
            shm_arena.forget(w.arena)
            # (the file stays while a pointer into it may still be waiting for its sender; the chunk directory is wiped at daemon start)
            w.arena.close(unlink=not any(o is not None and o.exists() for o in w._owner))
        self._tls.writer, self._tls.old_writers = None, []
        ls = getattr(self._tls, "link_slots", None)
        if ls is not None:
            ls.close()
            self._tls.link_slots = None
        rp = getattr(self._tls, "read_pool", None)
        if rp is not None:
            rp.shutdown(wait=False)
            self._tls.read_pool = None
        pp = getattr(self._tls, "publish_pool", None)
        if pp is not None:
            pp.shutdown(wait=True)
            self._tls.publish_pool = None
        if self._ctx is not None:
            self._arenas = {}             # the context frees its pinned blocks
            self._ctx.close()
            self._ctx = None


class GatewayHipDecompress(GatewayHipCompress):
    """op_type "gpu_decompress": the destination gateway's counterpart, in the place of the ``receive`` wait operator
    (GatewayWaitReceiver, skyplane/gateway/operators/gateway_operator.py:125-149).

    The reference's receiver decodes every chunk on the CPU inside recv_chunks (``lz4.frame.decompress``,
    gateway_receiver.py:195-201) and GatewayWaitReceiver polls until ``<id>.chunk`` has its full length.  With the
    receiver in deferred mode (hip_receiver.recv_chunks(decompress=None): the wire payload is left as
    ``<id>.chunk.lz4f``) this operator waits for the payloads instead, decodes whatever has arrived in ONE batched
    device call, checks what the reference checks and what it leaves as a todo --
      * decoded length == chunk_length_bytes      (gateway_receiver.py:213-218, GatewayWaitReceiver's assert :143-145)
      * MD5 of the decoded bytes == Chunk.md5_hash when the request carries one ("# todo check hash", :231); the digest
        is computed on the GPU right after the decode --
    writes ``<id>.chunk`` (temporary name + rename) and removes the payload.  A chunk whose payload has not arrived is
    re-queued exactly like GatewayWaitReceiver returns False; a chunk that arrived uncompressed (``<id>.chunk`` already
    complete) passes through.  A malformed frame or a digest mismatch raises -> error_queue + error_event, like the
    reference's ChecksumMismatchException path.  Shares the batching worker loop, device binding and pinned staging of
    GatewayHipCompress; there is no CPU fallback.
    """

    def __init__(self, *args, verify_md5: bool = True, dedup_wait_s: float = 60.0, dedup_store: str = "memory", out_slots: Optional[int] = None,
                 dedup_verify: str = "segments", **kwargs):
        lanes_given, wait_given = kwargs.get("pipeline_depth") is not None, kwargs.get("fill_wait_s") is not None
        super().__init__(*args, **kwargs)
        # How a chunk that travelled as a RECIPE is checked on the device path (round 6):
        #   "segments": every literal segment that arrives is digested where it was decoded (skyhip_segment_md5_device: thousands of independent messages at
        #       once, ~2.4 TB/s) and compared with the fingerprint its recipe carries; a reference names a segment that passed that test when IT arrived.  Every
        #       byte of the rebuilt chunk is then covered by an MD5 that matched, and no batch pays a serial whole-chunk chain (82 ms per 8 MiB chunk, whatever
        #       the batch holds: profiles/r5_dedup_wire.txt).  The sender's whole-chunk digest still travels with the request and is what the object store
        #       checks on upload (ContentMD5, skyplane/gateway/operators/gateway_operator.py:633-643).
        #   "chunk": round 5's rule -- the rebuilt chunk's own MD5 against Chunk.md5_hash, one chain per chunk and batch.
        assert dedup_verify in ("segments", "chunk")
        self.dedup_verify = dedup_verify
        # The base class's dedup rules are the SOURCE's (one lane: one fingerprint table; a short collect time for the lone lane).  A destination's lanes
        # share one segment store, and every one of its batches pays a whole-chunk digest chain whatever it holds: through the loopback a deduplicated
        # stream moves 18.9 Gbit/s with three lanes and 15.9 with two (profiles/r5_dedup_wire.txt), plain frames 44.0 with two and 38.1 with three.
        if not lanes_given:
            self.pipeline_depth = 3 if self.dedup_wire else 2
        if not wait_given:
            self.fill_wait_s = 0.03
        self.verify_md5 = verify_md5
        # raw side of the hand-off (round 5): decoded chunks are written by the device into page-locked slot FILES and published as hard links
        # (gateway/shm_arena.py::LinkSlots) instead of going pinned staging -> write() -> page cache.  Slots per lane: None = max_batch with the
        # "arena" hand-off (a batch in the making while the previous one's chunks are being uploaded and deleted), 0 with "files"; the slot size is the
        # chunk length of the lane's first batch (other lengths take the write path).
        self.out_slots = (self.max_batch if self.handoff == "arena" else 0) if out_slots is None else int(out_slots)
        self.out_slot_bytes = 1 << 30          # ... and at most this many bytes of slot files per lane
        assert dedup_store in ("memory", "files")
        self.dedup_store = dedup_store         # "files": the segment store lives in the chunk directory and several worker processes share it
        # dedup on the wire (dedup_wire.py): payloads that are recipes are rebuilt from their literal stream and the segments earlier chunks
        # brought.  The default segment store is shared by the lanes (threads) of ONE worker process: run this operator with n_processes=1 when the
        # source deduplicates, or with dedup_store="files" (FileSegmentStore, shared through the chunk directory) and as many as needed.  A chunk whose references cannot be resolved yet is re-queued; after dedup_wait_s it is an error.
        self.dedup_wait_s = float(dedup_wait_s)
        self._store = None                     # created in the worker process, on the first recipe
        self._store_lock = threading.Lock()
        self._first_miss = {}
        self._put_done = set()                 # chunk ids whose literal segments are in the store already (a retry must not store them again)

    def _process_in_lane(self, batch, worker_id):
        return self.process_batch(batch)

    def _segment_store(self, on_device: bool = False) -> "dedup_wire.SegmentStore":
        with self._store_lock:
            if self._store is None:
                if self.dedup_store == "files":
                    self._store = dedup_wire.FileSegmentStore(self.chunk_store.get_chunk_file_path("x").parent / "_segments")
                elif on_device:
                    self._store = dedup_wire.DeviceSegmentStore()      # fingerprint -> device address: the chunks are put together on the device
                else:
                    self._store = dedup_wire.SegmentStore()
            return self._store

    def process_exit(self, worker_id: int):
        super().process_exit(worker_id)
        if self._store is not None:            # the transfer is over for this process: what its segment store holds (RAM, or files in the chunk directory) goes
            self._store.cleanup()
            self._store = None

    @staticmethod
    def _expected_digest(chunk_req: ChunkRequest) -> Optional[bytes]:
        h = chunk_req.chunk.md5_hash
        if h is None:
            return None
        if isinstance(h, str):
            return bytes.fromhex(h)
        return bytes(h)

    def _rebuild_runs(self, cid: str, rec: "dedup_wire.Recipe", lit):
        """Device variant of _rebuild: (run addresses, run lengths, what must stay alive until the device has read them) -- DEVICE addresses of the byte runs
        that make up the chunk, neighbours merged -- or None
        while a referenced segment has not arrived.  lit = the chunk's decoded literal stream as a hip_ops.DeviceBuffer (None when it has none); the store
        (dedup_wire.DeviceSegmentStore) maps fingerprints to addresses inside such buffers: a later chunk's reference is a piece of an earlier chunk's
        stream, still where it was decoded.  Whole arrays in and out: no per-segment Python."""
        store = self._segment_store(on_device=True)
        segs = rec.segs
        nseg = len(segs)
        if not nseg:
            return np.zeros(0, np.uint64), np.zeros(0, np.uint32), None
        lens = segs["len"].astype(np.uint64)
        is_lit = segs["kind"] == dedup_wire.KIND_LITERAL
        llen = np.where(is_lit, lens, 0).astype(np.uint64)
        lit_start = np.cumsum(llen) - llen
        fp = np.ascontiguousarray(segs["fp"]).reshape(nseg, 16)
        li, ri = np.nonzero(is_lit)[0], np.nonzero(~is_lit)[0]
        src = np.zeros(nseg, np.uint64)
        keep = []
        if len(li):
            if lit is None:
                raise dedup_wire.RecipeError(f"chunk {cid}: literal segments without a literal stream")
            src[li] = np.uint64(lit.dptr) + lit_start[li]
            if cid not in self._put_done:
                store.put_arrays(rec.lane, rec.epoch, fp[li], src[li], lens[li].astype(np.uint32), lit)
                self._put_done.add(cid)
        if store.over_budget_live and not getattr(self, "_warned_over_budget", False):
            self._warned_over_budget = True      # said once: the budget never evicts what the sender may still reference, so the store outgrows it
            print(f"[{self.handle}] device segment store holds more than its byte budget in LIVE (lane, epoch) groups (bounded by lanes x keep_epochs x the "
                  "sender's dedup_epoch_bytes, counted as the device blocks they pin): raise the store's max_bytes or lower dedup_epoch_bytes", flush=True)
        if len(ri):
            addrs, hl, miss, keep = store.get_arrays(rec.lane, rec.epoch, fp[ri])
            if miss:
                t0 = self._first_miss.setdefault(cid, time.monotonic())
                if time.monotonic() - t0 > self.dedup_wait_s:
                    self._put_done.discard(cid)
                    k = int(ri[np.nonzero(addrs == 0)[0][0]])
                    raise ValueError(f"[Gateway] chunk {cid}: segment {fp[k].tobytes().hex()} of lane {rec.lane:#x} epoch {rec.epoch} did not arrive "
                                     f"within {self.dedup_wait_s:.0f} s (is gpu_decompress running with more than one worker process?)")
                return None
            bad = np.nonzero(hl.astype(np.uint64) != lens[ri])[0]
            if len(bad):
                k = int(ri[bad[0]])
                raise ValueError(f"[Gateway] chunk {cid}: referenced segment {fp[k].tobytes().hex()} has {int(hl[bad[0]])} bytes, the recipe says {int(lens[k])}")
            src[ri] = addrs
        self._first_miss.pop(cid, None)
        self._put_done.discard(cid)
        # neighbours in the chunk that are neighbours in memory are one run (a chunk's literals between two references; a run of references into one earlier stream)
        first = np.concatenate([[True], src[1:] != src[:-1] + lens[:-1]])
        starts = np.nonzero(first)[0]
        return src[starts], np.add.reduceat(lens, starts).astype(np.uint32), (lit, keep)      # (a run cannot exceed 32 bits: a chunk cannot -- max_chunk_bytes <= 1 GiB)

    def _verify_literal_segments(self, ctx, chunk_reqs, todo, recipes, datas, cached):
        """dedup_verify="segments": the literal segments of every recipe of the batch that is here for the first time -- ONE device call for all of them --
        against the fingerprints their recipes carry.  A recipe that waited (its literal stream is in the lane's cache) was checked when it first came."""
        addrs, lens, fps, owner = [], [], [], []
        for j, rec in enumerate(recipes):
            if rec is None or j in cached or not rec.lit_raw_len:
                continue
            lit = datas[j]
            if lit is None or len(lit) != rec.lit_raw_len:
                continue                               # (reported by the caller's length check)
            segs = rec.segs
            is_lit = segs["kind"] == dedup_wire.KIND_LITERAL
            ln = segs["len"][is_lit].astype(np.uint64)
            if not len(ln):
                continue
            if int(ln.max()) >= 1 << 15:
                raise ValueError(f"[Gateway] chunk {chunk_reqs[todo[j]].chunk.chunk_id}: a literal segment of {int(ln.max())} bytes (the CDC maximum is 16384)")
            addrs.append(np.uint64(lit.dptr) + (np.cumsum(ln) - ln)); lens.append(ln.astype(np.uint32))
            fps.append(np.ascontiguousarray(segs["fp"][is_lit]).reshape(-1, 16)); owner.append(np.full(len(ln), j))
        if not addrs:
            return
        got = ctx.segment_md5_device(np.concatenate(addrs), np.concatenate(lens))
        bad = np.nonzero((got != np.concatenate(fps)).any(axis=1))[0]
        if len(bad):
            j = int(np.concatenate(owner)[bad[0]])
            raise ValueError(f"[Gateway] chunk {chunk_reqs[todo[j]].chunk.chunk_id}: checksum mismatch, literal segment with md5 {got[bad[0]].tobytes().hex()} "
                             f"!= the recipe's fingerprint {np.concatenate(fps)[bad[0]].tobytes().hex()} ({len(bad)} of {len(got)} segments of the batch differ)")

    def _rebuild(self, cid: str, rec: "dedup_wire.Recipe", lit, out: Optional[np.ndarray] = None) -> Optional[np.ndarray]:
        """The chunk a recipe describes, or None while a referenced segment has not arrived.  lit = the decoded literal stream.  Runs of literal
        segments and runs of references into one earlier literal stream are each one copy; the only per-segment work is a dictionary look-up."""
        store = self._segment_store()
        segs = rec.segs
        nseg = len(segs)
        lens = segs["len"].astype(np.int64)
        is_lit = segs["kind"] == dedup_wire.KIND_LITERAL
        out_end = np.cumsum(lens)
        out_start = out_end - lens
        lit_end = np.cumsum(np.where(is_lit, lens, 0))
        lit_start = lit_end - np.where(is_lit, lens, 0)
        fpblob = segs["fp"].tobytes()
        lit = lit if isinstance(lit, np.ndarray) else np.frombuffer(lit, np.uint8)
        # this chunk's literals first: they may be what its own (or another waiting chunk's) references name
        li = np.nonzero(is_lit)[0]
        if len(li) and cid not in self._put_done:      # once per chunk: a chunk that waits for a reference comes back here many times
            store.put_chunk(rec.lane, rec.epoch, [fpblob[16 * k:16 * k + 16] for k in li], lit_start[li], lens[li], lit.tobytes())
            self._put_done.add(cid)
        if store.over_budget_live and not getattr(self, "_warned_over_budget", False):
            self._warned_over_budget = True      # said once: the budget never evicts what the sender may still reference, so the store outgrows it
            print(f"[{self.handle}] segment store holds more than its byte budget in LIVE (lane, epoch) groups (bounded by lanes x keep_epochs x the "
                  "sender's dedup_epoch_bytes): raise the store's max_bytes or lower dedup_epoch_bytes", flush=True)
        ri = np.nonzero(~is_lit)[0]
        hits = store.get_many(rec.lane, rec.epoch, [fpblob[16 * k:16 * k + 16] for k in ri]) if len(ri) else []
        for k, h in zip(ri, hits):
            if h is None:
                t0 = self._first_miss.setdefault(cid, time.monotonic())
                if time.monotonic() - t0 > self.dedup_wait_s:
                    self._put_done.discard(cid)
                    raise ValueError(f"[Gateway] chunk {cid}: segment {fpblob[16 * k:16 * k + 16].hex()} of lane {rec.lane:#x} epoch {rec.epoch} did not arrive "
                                     f"within {self.dedup_wait_s:.0f} s (is gpu_decompress running with more than one worker process?)")
                return None
            if h[2] != lens[k]:
                raise ValueError(f"[Gateway] chunk {cid}: referenced segment {fpblob[16 * k:16 * k + 16].hex()} has {h[2]} bytes, the recipe says {int(lens[k])}")
        out = np.empty(rec.raw_len, np.uint8) if out is None else out[:rec.raw_len]      # (a view of pinned staging when the caller has one: the digest call uploads from it)
        # literal runs: segments k..m literal <=> one contiguous piece of the literal stream
        if len(li):
            brk = np.nonzero(np.diff(li) != 1)[0]
            first, last = np.concatenate([[0], brk + 1]), np.concatenate([brk, [len(li) - 1]])
            for a, b in zip(li[first], li[last]):
                out[out_start[a]:out_end[b]] = lit[lit_start[a]:lit_end[b]]
        # reference runs: neighbours in the chunk that are neighbours in the same earlier literal stream
        j = 0
        while j < len(ri):
            buf, off, n = hits[j]
            k0, end = ri[j], off + n
            j += 1
            while j < len(ri) and ri[j] == ri[j - 1] + 1 and hits[j][0] is buf and hits[j][1] == end:
                end += hits[j][2]
                j += 1
            out[out_start[k0]:out_start[k0] + (end - off)] = np.frombuffer(buf, np.uint8, end - off, off)
        self._first_miss.pop(cid, None)
        self._put_done.discard(cid)
        return out

    def process_batch(self, chunk_reqs: List[ChunkRequest]) -> List[bool]:
        ctx = self._context()
        oks = [False] * len(chunk_reqs)
        self._last_metadata = [{} for _ in chunk_reqs]
        trace = [time.perf_counter()] if _OP_TRACE else None      # SKY_OP_TRACE=1: where a batch's time goes, one line per batch on stderr
        todo = []                                      # indices whose payload is there
        for i, cr in enumerate(chunk_reqs):
            cid = cr.chunk.chunk_id
            if sidecar.compressed_path(self.chunk_store, cid).exists():
                todo.append(i)
                continue
            raw = self.chunk_store.get_chunk_file_path(cid)
            if raw.exists():                           # sent uncompressed: GatewayWaitReceiver's test (:132-147)
                size = raw.stat().st_size
                if size >= cr.chunk.chunk_length_bytes:
                    assert size == cr.chunk.chunk_length_bytes, f"Downloaded chunk length does not match expected length: {size}, {cr.chunk.chunk_length_bytes}"
                    oks[i] = True
                    self._last_metadata[i] = {"uncompressed_size_bytes": size}
        if not todo:
            return oks
        pinned = hasattr(ctx, "pinned_buffer")
        paths = [sidecar.compressed_path(self.chunk_store, chunk_reqs[i].chunk.chunk_id) for i in todo]
        # a payload is a file, or a pointer to a slot of the receiver's shared arena (gateway/shm_arena.py): the latter is uploaded from where the
        # socket's bytes landed -- the arena is page-locked for this process on first sight -- instead of being read into a staging buffer first
        opened = [shm_arena.open_payload(p) for p in paths]
        sizes = [pl.length for pl in opened]
        chunk_lens = [int(chunk_reqs[i].chunk.chunk_length_bytes) for i in todo]
        payloads, into = [], None
        if pinned:
            arena = self._arena(ctx, "in", sum((pl.length + 255) & ~255 for pl in opened if pl.arena is None) or 1)
            out = self._arena(ctx, "out", sum((max(r, 1) + 255) & ~255 for r in chunk_lens))
            into, pi, po = [], 0, 0
            for p, pl, s, r in zip(paths, opened, sizes, chunk_lens):
                if pl.arena is not None:
                    shm_arena.register_once(pl.arena, ctx)
                    payloads.append(pl.view)
                else:
                    v = arena[pi:pi + s]
                    with open(p, "rb") as f:
                        got, mv = 0, memoryview(v)
                        while got < s:
                            k = f.readinto(mv[got:])
                            if not k:
                                break
                            got += k
                    assert got == s, f"payload {p.name} shrank while being read"
                    payloads.append(v)
                    pi += (s + 255) & ~255
                into.append(out[po:po + max(r, 1)])
                po += (max(r, 1) + 255) & ~255
        else:
            payloads = [pl.view if pl.arena is not None else p.read_bytes() for p, pl in zip(paths, opened)]
        # decoded chunks that fit a free slot file are written THERE by the device (and published as a hard link below): no write() of the chunk
        link = self._link_slots(ctx, chunk_lens)
        slot_of = {}
        if link is not None:
            fit = [j for j, r in enumerate(chunk_lens) if r == link.size]
            for j, sl in zip(fit, link.take(len(fit))):
                slot_of[j] = sl
        try:
            return self._decode_and_publish(ctx, chunk_reqs, todo, paths, payloads, sizes, chunk_lens, into, pinned, oks, trace, link, slot_of)
        except BaseException:
            for sl in slot_of.values():
                link.give_back(sl)
            raise

    def _link_slots(self, ctx, chunk_lens) -> Optional["shm_arena.LinkSlots"]:
        """This lane's slot files (made on first use, sized by the first batch's chunk length, page-locked through the lane's context)."""
        if self.out_slots <= 0 or not chunk_lens:
            return None
        ls = getattr(self._tls, "link_slots", None)
        # the slots are as long as the chunks of the transfer: the most frequent length of the batch at hand.  A lane whose first batch held only an
        # object's short tail is not stuck with that size (the arena's lesson): when a batch's usual length differs and no slot is in use, the
        # slot files are made again -- and that is said once per change
        lens_sorted = sorted(chunk_lens)
        size = max(set(lens_sorted), key=lens_sorted.count)
        if ls is not None and size > 0 and size != ls.size and ls.idle():
            print(f"[{self.handle}] slot files of {ls.size} bytes do not fit this transfer's chunks of {size} bytes: made again", flush=True)
            ls.close()
            ls = self._tls.link_slots = None
        if ls is None:
            if size <= 0:
                return None
            tag = f"{self.handle}_{os.getpid()}_{threading.get_ident() & 0xFFFFFF:x}"
            n_slots = max(2, min(self.out_slots, self.out_slot_bytes // size))      # (page-locked tmpfs: bounded in bytes too -- 64 MiB chunks get 16 slots per lane, not 64)
            # "slower, never wrong" (ADVICE r5): no room for the slot files on the chunk tmpfs (ENOSPC from posix_fallocate) or a page-lock that fails
            # (locked-memory limit) must not take the lane down -- the set is removed again, this lane stops asking, and its chunks take the write path
            ls = None
            try:
                ls = shm_arena.LinkSlots(self.chunk_store.get_chunk_file_path("x").parent, tag, size, n_slots)
                ls.register(ctx)
            except (OSError, MemoryError, RuntimeError) as e:
                if ls is not None:
                    ls.close()
                self.out_slots = 0
                print(f"[{self.handle}] no page-locked slot files for decoded chunks ({n_slots} x {size} bytes: {type(e).__name__}: {e}): chunks are written "
                      "through write() instead", flush=True)
                return None
            self._tls.link_slots = ls
        return ls

    def _decode_and_publish(self, ctx, chunk_reqs, todo, paths, payloads, sizes, chunk_lens, into, pinned, oks, trace, link, slot_of) -> List[bool]:
        # a payload is an LZ4 frame of the chunk, or a recipe whose literal stream is one (dedup_wire.py): one batched decode for both kinds
        recipes = [None] * len(todo)
        frames, raw_lens = list(payloads), list(chunk_lens)
        for j, (pl, r) in enumerate(zip(payloads, chunk_lens)):
            if dedup_wire.is_recipe(pl):
                rec = dedup_wire.parse_recipe(memoryview(pl), max_raw_len=r)
                if rec.raw_len != r:
                    raise ValueError(f"[Gateway] chunk {chunk_reqs[todo[j]].chunk.chunk_id}: recipe for {rec.raw_len} bytes, expected {r}")
                recipes[j] = rec
                frames[j], raw_lens[j] = rec.lit_frame, rec.lit_raw_len
        # a recipe that was here before and had to wait left its decoded literal stream behind: no second decode for it
        cache = self._tls.__dict__.setdefault("lit_cache", {})
        cached = {j: cache[chunk_reqs[todo[j]].chunk.chunk_id] for j in range(len(todo)) if recipes[j] is not None and chunk_reqs[todo[j]].chunk.chunk_id in cache}
        dec = [j for j in range(len(todo)) if (recipes[j] is None or recipes[j].lit_raw_len) and j not in cached]       # (a recipe of references only has nothing to decode)
        want = self.verify_md5 and any(self._expected_digest(chunk_reqs[i]) is not None for i in todo)
        want_dec = want and any(recipes[j] is None for j in dec)      # the digest of a literal stream is of no use (and costs a whole MD5 chain)
        kwargs = {"want_md5": True} if want_dec else {}
        if into is not None:
            # (a frame decodes into its slot file's pages; a recipe's decode yields its literal stream, which goes to staging -- the REBUILT chunk goes to the slot)
            for j in dec:
                if recipes[j] is None and j in slot_of:
                    into[j] = link.views[slot_of[j]]
            kwargs["into"] = [into[j] for j in dec]
        datas, digests = [np.zeros(0, np.uint8)] * len(todo), [None] * len(todo)
        if trace:
            trace.append(time.perf_counter())
        # Round 5: with a real device behind the context (and the in-process segment store) a recipe's literal stream is decoded into DEVICE memory and stays
        # there -- the store keeps device buffers --, and the chunk is put together on the device from runs of this and earlier streams, digested there and
        # copied out once (skyhip_decompress_to_device / skyhip_gather_md5).  Rounds 2-4: literal stream to the host, numpy copies, chunk uploaded again.
        on_device = hasattr(ctx, "gather_md5") and hasattr(ctx, "decompress_to_device") and self.dedup_store == "memory" and any(r is not None for r in recipes)
        if on_device:
            dec_dev = [j for j in dec if recipes[j] is not None]
            dec = [j for j in dec if recipes[j] is None]
            if "into" in kwargs:
                kwargs["into"] = [into[j] for j in dec]
            if dec_dev:
                try:
                    bufs = ctx.decompress_to_device([frames[j] for j in dec_dev], [raw_lens[j] for j in dec_dev])
                except Exception as e:      # noqa: BLE001
                    if getattr(e, "code", None) != -2:       # SKYHIP_E_NOMEM (include/skyhip.h)
                        raise
                    # device memory is what the segment store spends: let go of every group the sender can no longer reference and try once more
                    import gc

                    dropped = self._segment_store(on_device=True).drop_not_live()
                    gc.collect()
                    print(f"[{self.handle}] out of device memory for {len(dec_dev)} literal streams: dropped {dropped} retired segment groups, trying again", flush=True)
                    bufs = ctx.decompress_to_device([frames[j] for j in dec_dev], [raw_lens[j] for j in dec_dev])
                for j, buf in zip(dec_dev, bufs):
                    datas[j] = buf
            for j in range(len(todo)):
                if recipes[j] is not None and j not in cached and not recipes[j].lit_raw_len:
                    datas[j] = None                    # (nothing but references)
        if dec:
            res = ctx.decompress_batch([frames[j] for j in dec], [raw_lens[j] for j in dec], **kwargs)
            dd, gg = res if want_dec else (res, [None] * len(dec))
            for j, d, g in zip(dec, dd, gg):
                datas[j], digests[j] = d, g
        for j, lit in cached.items():
            datas[j] = lit
        if trace:
            trace.append(time.perf_counter())
        # recipes: rebuild; their digests are those of the rebuilt chunks (one more device call, MD5 only)
        ready = [True] * len(todo)
        rebuilt = []
        n_rec_bytes = sum((recipes[j].raw_len + 255) & ~255 for j in range(len(todo)) if recipes[j] is not None)
        reb_arena = self._arena(ctx, "rebuilt", n_rec_bytes) if (pinned and n_rec_bytes) else None
        reb_pos = 0
        seg_verified = set()
        if on_device:
            gj, g_src, g_len, g_into, g_keep = [], [], [], [], []
            by_segment = self.verify_md5 and self.dedup_verify == "segments" and hasattr(ctx, "segment_md5_device")
            if by_segment:
                self._verify_literal_segments(ctx, chunk_reqs, todo, recipes, datas, cached)
            for j, rec in enumerate(recipes):
                if rec is None:
                    continue
                lit = datas[j]
                if (0 if lit is None else len(lit)) != rec.lit_raw_len:
                    raise ValueError(f"[Gateway] chunk {chunk_reqs[todo[j]].chunk.chunk_id}: literal stream of {0 if lit is None else len(lit)} bytes, the recipe says {rec.lit_raw_len}")
                cid_j = chunk_reqs[todo[j]].chunk.chunk_id
                if by_segment:
                    seg_verified.add(j)
                try:
                    runs = self._rebuild_runs(cid_j, rec, lit)
                except BaseException:
                    cache.pop(cid_j, None)
                    raise
                if runs is None:
                    ready[j] = False
                    if cid_j not in cache and len(cache) < 4 * self.max_batch:
                        cache[cid_j] = lit             # (a device buffer: it stays where it is)
                    continue
                cache.pop(cid_j, None)
                if j in slot_of:
                    dstv = link.views[slot_of[j]]
                else:
                    dstv = reb_arena[reb_pos:reb_pos + rec.raw_len]
                    reb_pos += (rec.raw_len + 255) & ~255
                gj.append(j); g_src.append(runs[0]); g_len.append(runs[1]); g_into.append(dstv); g_keep.append(runs[2])
            if gj:
                outs, digs = ctx.gather_md5(g_src, g_len, g_into, want_md5=bool(want) and not by_segment)
                g_keep.clear()                         # the device has read every run: the buffers may go when their groups do
                for k, j in enumerate(gj):
                    datas[j], digests[j] = outs[k], (digs[k] if digs is not None else None)
            recipes_done = True
        else:
            recipes_done = False
        for j, rec in enumerate(recipes):
            if rec is None or recipes_done:
                continue
            if len(datas[j]) != rec.lit_raw_len:
                raise ValueError(f"[Gateway] chunk {chunk_reqs[todo[j]].chunk.chunk_id}: literal stream of {len(datas[j])} bytes, the recipe says {rec.lit_raw_len}")
            cid_j = chunk_reqs[todo[j]].chunk.chunk_id
            try:
                slot = None
                if j in slot_of:
                    slot = link.views[slot_of[j]]
                elif reb_arena is not None:
                    slot = reb_arena[reb_pos:reb_pos + rec.raw_len]
                    reb_pos += (rec.raw_len + 255) & ~255
                chunk = self._rebuild(cid_j, rec, datas[j], out=slot)
            except BaseException:
                cache.pop(cid_j, None)
                raise
            if chunk is None:
                ready[j] = False
                if cid_j not in cache and len(cache) < 4 * self.max_batch:
                    d = datas[j]       # (bytes from a plain context, or a view of the staging arena that the next call reuses: keep a copy)
                    cache[cid_j] = np.array(d, dtype=np.uint8, copy=True) if isinstance(d, np.ndarray) else np.frombuffer(bytes(d), np.uint8)
                continue
            cache.pop(cid_j, None)
            datas[j], digests[j] = chunk, None
            rebuilt.append(j)
        if want and rebuilt:
            for j, r in zip(rebuilt, ctx.process_batch([datas[j] for j in rebuilt], flags=2)):
                digests[j] = r.md5
        if trace:
            trace.append(time.perf_counter())
        for j, (i, p, data, dig, size) in enumerate(zip(todo, paths, datas, digests, sizes)):
            if not ready[j]:
                if j in slot_of:
                    link.give_back(slot_of.pop(j))
                continue                               # re-queued by the worker loop; its literals are already in the store
            cr = chunk_reqs[i]
            cid = cr.chunk.chunk_id
            if len(data) != cr.chunk.chunk_length_bytes:
                raise ValueError(f"[Gateway] chunk {cid}: {len(data)} bytes after decoding, expected {cr.chunk.chunk_length_bytes}")
            exp = self._expected_digest(cr) if self.verify_md5 else None
            if j in seg_verified:
                exp = None                             # checked segment by segment above (dedup_verify="segments")
            if exp is not None and dig != exp:
                raise ValueError(f"[Gateway] chunk {cid}: checksum mismatch, md5 {dig.hex()} != {exp.hex()}")
            final = self.chunk_store.get_chunk_file_path(cid)
            sl = slot_of.pop(j, None)
            if sl is not None and isinstance(data, np.ndarray) and data.size == link.size and data.ctypes.data == link.views[sl].ctypes.data:
                link.publish(sl, final)                # the chunk's bytes are already in the file's pages: a hard link makes them <id>.chunk
            else:
                if sl is not None:                     # (a context that does not decode in place -- emulator -- : one copy into the slot's pages)
                    link.views[sl][:len(data)] = np.frombuffer(data, np.uint8) if not isinstance(data, np.ndarray) else data
                    link.publish(sl, final)
                else:
                    tmp = final.with_suffix(".dectmp")
                    with open(tmp, "wb") as f:
                        f.write(data)
                    os.replace(tmp, final)
            p.unlink()
            meta = {"compressed_size_bytes": size, "uncompressed_size_bytes": len(data)}
            if dig is not None:
                meta["md5_hex"] = dig.hex()
            if recipes[j] is not None:
                meta["dedup_reference_bytes"] = int(recipes[j].raw_len - recipes[j].lit_raw_len)
                if j in seg_verified:
                    meta["verified"] = "segment fingerprints"
            self._last_metadata[i] = meta
            oks[i] = True
        if trace:
            t = trace + [time.perf_counter()]
            print(f"[op-trace] {self.handle} pid {os.getpid()} {threading.current_thread().name}: {len(todo)} chunks at {t[0]:.3f}: open+stage {1e3 * (t[1] - t[0]):.1f} ms, "
                  f"device {1e3 * (t[2] - t[1]):.1f}, rebuild+digest {1e3 * (t[3] - t[2]):.1f}, write {1e3 * (t[4] - t[3]):.1f}", file=sys.stderr, flush=True)
        return oks
