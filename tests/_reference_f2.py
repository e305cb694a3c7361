"""SURVEY 8f item 2, the SOURCE's raw side, as an EXECUTABLE patch against the reference's own source (build container only; run by
tests/test_reference_interop.py).  INTEGRATION.md section 6e, applied IN MEMORY (nothing from /root/reference is copied into the repo):

  a. `download_object` opens an EXISTING destination without truncating it (skyplane/obj_store/posix_file_interface.py:105,109 here; the same two-word
     edit in s3_interface.py:183 and its siblings) -- a truncation would free the pages gpu_compress page-locked;
  b. GatewayObjStoreReadOperator.process (skyplane/gateway/operators/gateway_operator.py:531-575) makes `<id>.chunk` a hard link to a free source slot
     before it downloads (skyplane_amd.gateway.shm_arena.claim_slot: one line, a no-op when gpu_compress offers no slots).

The patched reader then downloads three chunks of a POSIX "bucket" object: two of the transfer's chunk size land in gpu_compress's slot files (the
operator consumes them where they lie: no read of the file), the object's short tail stays an ordinary file; everything decodes to the object's bytes; a
deleted chunk frees its slot for the next claim.  TEST INFRASTRUCTURE ONLY.
"""
import hashlib
import os
import sys
import tempfile
import types
import uuid
from multiprocessing import Event, Queue
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from oracle import ref, refshim  # noqa: E402

scratch = Path(tempfile.mkdtemp(prefix="sky_f2_"))
refshim.install(scratch / "shim")
REF = refshim.REFERENCE / "skyplane"

import skyplane.chunk as ref_chunk  # noqa: E402
import skyplane.gateway.chunk_store as ref_chunk_store  # noqa: E402
import skyplane.gateway.gateway_queue as ref_queue  # noqa: E402

sys.modules["skyplane_amd.chunk"] = ref_chunk
sys.modules["skyplane_amd.gateway.chunk_store"] = ref_chunk_store
sys.modules["skyplane_amd.gateway.gateway_queue"] = ref_queue
from skyplane_amd import synth  # noqa: E402
from skyplane_amd.gateway import shm_arena, sidecar  # noqa: E402
from skyplane_amd.gateway.operators.gateway_operator import GatewayHipCompress  # noqa: E402
from tests._reference_dropin import EmuContext  # noqa: E402  (the shipping kernels under the CPU emulator; importing it also builds the emulator)

POSIX_EDIT = ('with open(dst_file_path, "wb") as dst_file:', 'with open(dst_file_path, "r+b" if os.path.exists(dst_file_path) else "wb") as dst_file:')
READ_OP_EDIT = ('            # create empty file\n            Path(fpath).touch()\n            return True\n',
                '            # create empty file\n            Path(fpath).touch()\n            return True\n'
                '        claim_slot(fpath, chunk_req.chunk.chunk_length_bytes)      # gpu_compress\'s page-locked source slots (no-op when there are none)\n')


def _load(modname, path, src, inject=None):
    mod = types.ModuleType(modname)
    mod.__file__ = str(path)
    if inject:
        mod.__dict__.update(inject)
    exec(compile(src, str(path), "exec"), mod.__dict__)
    return mod


def main():
    # ---- the two edits ----
    pp = REF / "obj_store" / "posix_file_interface.py"
    ptxt = pp.read_text()
    assert ptxt.count(POSIX_EDIT[0]) == 2, "INTEGRATION 6e anchor (posix download_object) not found exactly twice"
    posix_mod = _load("skyplane.obj_store.posix_file_interface_f2", pp, ptxt.replace(POSIX_EDIT[0], POSIX_EDIT[1]))
    op_path = REF / "gateway" / "operators" / "gateway_operator.py"
    otxt = op_path.read_text()
    assert otxt.count(READ_OP_EDIT[0]) == 1, "INTEGRATION 6e anchor (GatewayObjStoreReadOperator.process) not found exactly once"
    op_mod = _load("skyplane.gateway.operators.gateway_operator_f2", op_path, otxt.replace(READ_OP_EDIT[0], READ_OP_EDIT[1]), inject={"claim_slot": shm_arena.claim_slot})

    # ---- a POSIX "bucket" object of two full chunks and a tail ----
    cs = 192 * 1024
    obj = synth.silesia_like(2 * cs + 50_001, config_id=6, seg_min=3000, seg_max=40000).tobytes()
    obj_path = scratch / "bucket_object.bin"
    obj_path.write_bytes(obj)
    store = ref_chunk_store.ChunkStore(str(scratch / "src_chunks"))
    q_in, q_out = ref_queue.GatewayQueue(), ref_queue.GatewayQueue()
    store.add_partition("0", q_in)
    comp = GatewayHipCompress("gpu_compress_0", "local:src", q_in, q_out, Event(), Queue(), store, n_processes=1, max_batch=4, max_chunk_bytes=cs, device_ids=[0],
                              context_factory=lambda d, mc, mb: EmuContext(d, mc, mb), in_slots=2, in_slot_chunk_bytes=cs, handoff="files")
    ctx = comp._context()
    assert comp._in_slots(ctx, [cs]) is not None                      # (what the lane's prealloc does at start)
    slot_files = sorted(Path(store.get_chunk_file_path("x")).parent.glob("_inslot_*"))
    assert len(slot_files) == 2 and all(f.stat().st_size == cs and f.stat().st_nlink == 1 for f in slot_files)

    reader = op_mod.GatewayObjStoreReadOperator("read", "local:src", "bucket", "local:src", ref_queue.GatewayQueue(), ref_queue.GatewayQueue(), Event(), Queue(),
                                                n_processes=1, chunk_store=store)
    reader.worker_id = 0
    reader.obj_store_interfaces["local:src:bucket"] = posix_mod.POSIXInterface()
    reqs = []
    for k, (off, ln) in enumerate([(0, cs), (cs, cs), (2 * cs, len(obj) - 2 * cs)]):
        cid = uuid.uuid4().hex
        cr = ref_chunk.ChunkRequest(chunk=ref_chunk.Chunk(src_key=str(obj_path), dest_key=f"d{k}", chunk_id=cid, chunk_length_bytes=ln, file_offset_bytes=off, partition_id="0"))
        assert reader.process(cr) is True                             # the reference's own download path, patched
        reqs.append(cr)
    inodes = {f.stat().st_ino for f in slot_files}
    st = [os.stat(store.get_chunk_file_path(cr.chunk.chunk_id)) for cr in reqs]
    assert [s.st_ino in inodes for s in st] == [True, True, False] and [s.st_nlink for s in st] == [2, 2, 1], "two chunks in slots, the tail an ordinary file"
    for cr, (off, ln) in zip(reqs, [(0, cs), (cs, cs), (2 * cs, len(obj) - 2 * cs)]):
        assert store.get_chunk_file_path(cr.chunk.chunk_id).read_bytes() == obj[off:off + ln]

    # ---- gpu_compress consumes the slot-resident chunks where they lie ----
    opened = []
    real_open = open

    def spy(path, *a, **kw):
        opened.append(str(path))
        return real_open(path, *a, **kw)

    import builtins
    builtins.open = spy
    try:
        oks = comp.process_batch(reqs)
    finally:
        builtins.open = real_open
    assert oks == [True] * 3 and comp._tls.in_slot_hits == 2
    raw_opens = [p for p in opened if p.endswith(".chunk")]
    assert raw_opens == [str(store.get_chunk_file_path(reqs[2].chunk.chunk_id))], f"only the tail's file is read: {raw_opens}"
    for cr, (off, ln) in zip(reqs, [(0, cs), (cs, cs), (2 * cs, len(obj) - 2 * cs)]):
        frame = sidecar.compressed_path(store, cr.chunk.chunk_id).read_bytes()
        assert ref.lz4f_decompress(frame, ln) == obj[off:off + ln]
        assert sidecar.digest_path(store, cr.chunk.chunk_id).read_text() == hashlib.md5(obj[off:off + ln]).hexdigest()
    # ---- the daemon's unlink of <id>.chunk frees the slot; no free slot -> an ordinary file ----
    cid = uuid.uuid4().hex
    cr = ref_chunk.ChunkRequest(chunk=ref_chunk.Chunk(src_key=str(obj_path), dest_key="d3", chunk_id=cid, chunk_length_bytes=cs, file_offset_bytes=7, partition_id="0"))
    assert reader.process(cr) is True and os.stat(store.get_chunk_file_path(cid)).st_nlink == 1      # both slots are still linked
    os.unlink(store.get_chunk_file_path(cid))
    os.unlink(store.get_chunk_file_path(reqs[0].chunk.chunk_id))
    shm_arena._claim_cache.clear()
    assert reader.process(cr) is True and os.stat(store.get_chunk_file_path(cid)).st_ino in inodes
    assert store.get_chunk_file_path(cid).read_bytes() == obj[7:7 + cs]
    comp.worker_exit(0)
    comp.process_exit(0)
    assert not list(Path(store.get_chunk_file_path("x")).parent.glob("_inslot_*"))
    print("OK f2 source slots: 2 chunks consumed in place, tail read, slot reuse")


if __name__ == "__main__":
    main()
