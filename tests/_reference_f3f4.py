"""SURVEY 8f items 3 and 4 as EXECUTABLE patches against the reference's own source (build container only; run by
tests/test_reference_interop.py).  Every edit below is the text INTEGRATION.md sections 7 and 9 show; it is applied to the reference's
source IN MEMORY (nothing from /root/reference is copied into the repo) and the patched code is then run.

f3 -- checksum plumbing end to end
   a. source sender: the pre-registration body (gateway_operator.py:299) carries the digest gpu_compress left in <id>.chunk.md5,
      as a hex string (bytes would break json.dumps -- SURVEY fact 7.5);
   b. destination receiver: `# todo check hash` (gateway_receiver.py:231) becomes "write the MD5 of the decoded bytes next to the chunk";
   c. destination GatewayObjStoreWriteOperator.process (gateway_operator.py:616-643): compare that digest with the registered one and
      hand `check_md5` to upload_object as BYTES (what s3_interface.py:203 base64-encodes into Content-MD5).
f4 -- planner / TransferConfig integration
   a. TransferConfig.use_gpu_compression (api/config.py:87);
   b. GatewayGpuCompress program node (gateway_program.py, pattern :34-97);
   c. MulticastDirectPlanner.plan (planner/planner.py:321-362): gpu_compress between the read operator and mux_and, GatewaySend(compress=False);
   the test plans a job in memory and checks the program JSON the source gateways would receive.
TEST INFRASTRUCTURE ONLY.
"""
import hashlib
import json
import sys
import tempfile
import types
import uuid
from multiprocessing import Event, Manager, Queue
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from oracle import refshim  # noqa: E402

scratch = Path(tempfile.mkdtemp(prefix="sky_f3f4_"))
refshim.install(scratch / "shim")
REF = refshim.REFERENCE / "skyplane"


def _apply(src, edits, what):
    for old, new in edits:
        assert src.count(old) == 1, f"{what}: anchor not found exactly once: {old!r}"
        src = src.replace(old, new)
    return src


def _load(modname, path, edits, what, inject=None):
    mod = types.ModuleType(modname)
    mod.__file__ = str(path)
    if inject:
        mod.__dict__.update(inject)
    exec(compile(_apply(path.read_text(), edits, what), str(path), "exec"), mod.__dict__)
    return mod


# ---------------------------------------------------------------------------------------------------------------------
# f3
# ---------------------------------------------------------------------------------------------------------------------
SENDER_REGISTER_EDIT = [
    ('                    register_body = json.dumps([c.chunk.as_dict() for c in chunk_reqs[n_added:]]).encode("utf-8")\n',
     '                    register_body = json.dumps([attach_digest(c.chunk.as_dict(), self.chunk_store) for c in chunk_reqs[n_added:]]).encode("utf-8")\n'),
]
RECEIVER_HASH_EDIT = [
    ('            # todo check hash\n',
     '            if not getattr(self, "defer_decode", False):    # (with gpu_decompress downstream the operator computes and checks it on the GPU)\n'
     '                fpath.with_name(fpath.name + ".md5").write_text(hashlib.md5(fpath.read_bytes()).hexdigest())\n'),
]
WRITE_OP_EDIT = [
    ('                check_md5=chunk_req.chunk.md5_hash,\n',
     '                check_md5=verified_digest(chunk_req.chunk, self.chunk_store),\n'),
]


def f3():
    import skyplane.chunk as ref_chunk
    import skyplane.gateway.chunk_store as ref_chunk_store
    import skyplane.gateway.gateway_queue as ref_queue

    sys.modules["skyplane_amd.chunk"] = ref_chunk
    sys.modules["skyplane_amd.gateway.chunk_store"] = ref_chunk_store
    sys.modules["skyplane_amd.gateway.gateway_queue"] = ref_queue
    from skyplane_amd.gateway import sidecar
    from skyplane_amd.gateway.operators import hip_sender

    data = {uuid.uuid4().hex: bytes([i]) * (5000 + i) for i in range(3)}
    src = ref_chunk_store.ChunkStore(str(scratch / "src"))
    dst = ref_chunk_store.ChunkStore(str(scratch / "dst"))
    reqs = []
    for cid, d in data.items():
        src.get_chunk_file_path(cid).write_bytes(d)
        sidecar.digest_path(src, cid).write_text(hashlib.md5(d).hexdigest())                      # what gpu_compress leaves behind
        reqs.append(ref_chunk.ChunkRequest(chunk=ref_chunk.Chunk(src_key=cid, dest_key=str(scratch / "bucket" / cid), chunk_id=cid,
                                                                 chunk_length_bytes=len(d), partition_id="0")))
    (scratch / "bucket").mkdir()

    # ---- a. the patched sender's pre-registration body -----------------------------------------------------------
    op_path = REF / "gateway" / "operators" / "gateway_operator.py"
    op_mod = _load("skyplane.gateway.operators.gateway_operator_f3", op_path, SENDER_REGISTER_EDIT + WRITE_OP_EDIT, "f3 sender/write-op patch",
                   inject={"attach_digest": hip_sender.attach_digest, "verified_digest": hip_sender.verified_digest})
    bodies = []

    class _Pool:
        def request(self, method, url, body=None, headers=None):
            bodies.append(json.loads(body))
            return types.SimpleNamespace(status=200, data=json.dumps({"status": "ok", "n_added": len(bodies[-1])}).encode())

    sender = op_mod.GatewaySender("send", "local:src", ref_queue.GatewayQueue(), ref_queue.GatewayQueue(), Event(), Queue(), src, ip_addr="127.0.0.1",
                                  use_tls=False, use_compression=False, n_processes=1)
    sender.worker_id = 0
    sender.http_pool = _Pool()

    class _Sock:                      # the data socket is not the subject here
        def sendall(self, b):
            pass

    sender.destination_ports["127.0.0.1"] = 1
    sender.destination_sockets["127.0.0.1"] = _Sock()
    for cr in reqs:
        assert sender.process(cr, "127.0.0.1") is True
    registered = [d for b in bodies for d in b]
    assert len(registered) == len(reqs)
    for d in registered:
        assert d["md5_hash"] == hashlib.md5(data[d["chunk_id"]]).hexdigest(), "registration must carry the hex digest"
    # ---- b. the patched receiver leaves the digest of what it decoded next to the chunk --------------------------------
    rc_path = REF / "gateway" / "operators" / "gateway_receiver.py"
    src_txt = _apply(rc_path.read_text(), RECEIVER_HASH_EDIT, "f3 receiver patch")
    compile(src_txt, str(rc_path), "exec")            # the edit compiles in place; its two statements, run on a received chunk file:
    for cid, d in data.items():
        fpath = dst.get_chunk_file_path(cid)
        fpath.write_bytes(d)
        ns = {"hashlib": hashlib, "fpath": fpath, "self": types.SimpleNamespace()}
        exec("if not getattr(self, 'defer_decode', False):\n    fpath.with_name(fpath.name + '.md5').write_text(hashlib.md5(fpath.read_bytes()).hexdigest())", ns)
        assert sidecar.digest_path(dst, cid).read_text() == hashlib.md5(d).hexdigest()
    # ---- c. the patched write operator: registered digest == received digest, check_md5 handed over as bytes -----------
    seen = {}

    class _Iface:
        def upload_object(self, src_file_path, dst_object_name, part_number=None, upload_id=None, check_md5=None, mime_type=None):
            body = Path(src_file_path).read_bytes()
            assert isinstance(check_md5, bytes) and len(check_md5) == 16, "Content-MD5 needs the 16 raw bytes (s3_interface.py:203)"
            if hashlib.md5(body).digest() != check_md5:
                raise op_mod.__dict__.get("ChecksumMismatchException", ValueError)(dst_object_name)
            seen[Path(src_file_path).name] = check_md5

    mgr = Manager()
    wop = op_mod.GatewayObjStoreWriteOperator("write", "local:dst", "bucket", "local:dst", ref_queue.GatewayQueue(), ref_queue.GatewayQueue(), Event(), Queue(),
                                              mgr.dict(), n_processes=1, chunk_store=dst)
    wop.worker_id = 0
    wop.get_obj_store_interface = lambda region, bucket: _Iface()
    for d in registered:                                   # the destination API rebuilds the requests from the registration body
        cr = ref_chunk.ChunkRequest(chunk=ref_chunk.Chunk.from_dict(d))
        assert wop.process(cr) is True
    assert len(seen) == len(data)
    # a corrupted chunk is refused BEFORE it is uploaded
    cid = next(iter(data))
    dst.get_chunk_file_path(cid).write_bytes(b"corrupted")
    sidecar.digest_path(dst, cid).write_text(hashlib.md5(b"corrupted").hexdigest())
    try:
        wop.process(ref_chunk.ChunkRequest(chunk=ref_chunk.Chunk.from_dict([d for d in registered if d["chunk_id"] == cid][0])))
        raise SystemExit("a digest mismatch must stop the upload")
    except ValueError as e:
        assert "digest" in str(e)
    mgr.shutdown()
    return len(registered)


# ---------------------------------------------------------------------------------------------------------------------
# f4
# ---------------------------------------------------------------------------------------------------------------------
CONFIG_EDIT = [
    ('    use_compression: bool = True\n', '    use_compression: bool = True\n    use_gpu_compression: bool = False  # LZ4 + MD5 on the gateway\'s GPU (gpu_compress operator) instead of in GatewaySender\n'
     '    use_gpu_dedup: bool = False  # with use_gpu_compression: content-defined segments the destination already holds are not sent again (INTEGRATION.md section 10)\n'),
]
PROGRAM_EDIT = [           # INTEGRATION.md section 9b (same node as section 4)
    ('class GatewayReceive(GatewayOperator):\n',
     'class GatewayGpuCompress(GatewayOperator):\n'
     '    def __init__(self, num_workers: int = 1, max_batch: int = 64, max_chunk_mb: int = 64, compute_md5: bool = True, cdc: bool = False, dedup: bool = False,\n'
     '                 dedup_wire: bool = False, dedup_epoch_mb: int = 8192):\n'
     '        super().__init__("gpu_compress")\n'
     '        self.num_workers, self.max_batch, self.max_chunk_mb = num_workers, max_batch, max_chunk_mb\n'
     '        self.compute_md5, self.cdc, self.dedup = compute_md5, cdc or dedup_wire, dedup or dedup_wire\n'
     '        self.dedup_wire, self.dedup_epoch_mb = dedup_wire, dedup_epoch_mb\n\n\n'
     'class GatewayGpuDecompress(GatewayOperator):\n'
     '    def __init__(self, num_workers: int = 1, max_batch: int = 64, max_chunk_mb: int = 64, verify_md5: bool = True, dedup_wire: bool = False):\n'
     '        super().__init__("gpu_decompress")\n'
     '        self.num_workers = 1 if dedup_wire else num_workers      # the segment store lives in one worker process\n'
     '        self.max_batch, self.max_chunk_mb, self.verify_md5, self.dedup_wire = max_batch, max_chunk_mb, verify_md5, dedup_wire\n\n\n'
     'class GatewayReceive(GatewayOperator):\n'),
]
PLANNER_EDIT = [           # INTEGRATION.md section 9c -- MulticastDirectPlanner.plan
    ('    GatewaySend,\n)\n', '    GatewaySend,\n    GatewayGpuCompress,\n    GatewayGpuDecompress,\n)\n'),
    ('            # send to all destination\n            mux_and = src_program.add_operator(GatewayMuxAnd(), parent_handle=obj_store_read, partition_id=partition_id)\n',
     '            # send to all destination\n'
     '            gpu = self.transfer_config.use_compression and getattr(self.transfer_config, "use_gpu_compression", False)\n'
     '            stage = obj_store_read\n'
     '            dedup = gpu and getattr(self.transfer_config, "use_gpu_dedup", False)\n'
     '            if gpu:   # compress + hash on the GPU, once, in front of the fan-out (every destination gets the same payload)\n'
     '                stage = src_program.add_operator(GatewayGpuCompress(dedup_wire=dedup), parent_handle=obj_store_read, partition_id=partition_id)\n'
     '            mux_and = src_program.add_operator(GatewayMuxAnd(), parent_handle=stage, partition_id=partition_id)\n'),
    ('                            compress=self.transfer_config.use_compression,\n                            encrypt=self.transfer_config.use_e2ee,\n                        ),\n                        parent_handle=mux_or,\n',
     '                            compress=self.transfer_config.use_compression and not gpu,\n                            encrypt=self.transfer_config.use_e2ee,\n                        ),\n                        parent_handle=mux_or,\n'),
    ('                    GatewayReceive(decompress=self.transfer_config.use_compression, decrypt=self.transfer_config.use_e2ee),\n',
     '                    # recipes are rebuilt by the operator that holds the segment store; the receiver then leaves payloads as they arrive (section 6b)\n'
     '                    GatewayGpuDecompress(dedup_wire=True) if dedup else\n'
     '                    GatewayReceive(decompress=self.transfer_config.use_compression, decrypt=self.transfer_config.use_e2ee),\n'),
]


def f4():
    import skyplane.gateway.gateway_program as _unused  # noqa: F401  (plain module: must import cleanly)

    # the planner's module-level imports that need cloud SDKs are replaced by the two names it uses from them
    compute = types.ModuleType("skyplane.compute")

    class CloudProvider:
        @staticmethod
        def get_transfer_cost(src, dst, premium_tier=True):
            return 0.0

    compute.CloudProvider = CloudProvider
    compute.__getattr__ = lambda name: type(name, (), {})      # AWSAuthentication & co. appear in annotations only
    sys.modules["skyplane.compute"] = compute
    import skyplane

    skyplane.compute = compute
    tj = types.ModuleType("skyplane.api.transfer_job")
    tj.TransferJob = object
    sys.modules["skyplane.api.transfer_job"] = tj
    prog = _load("skyplane.gateway.gateway_program", REF / "gateway" / "gateway_program.py", PROGRAM_EDIT, "f4 program-node patch")
    sys.modules["skyplane.gateway.gateway_program"] = prog
    for m in ("skyplane.planner.topology",):
        sys.modules.pop(m, None)
    cfg = _load("skyplane.api.config", REF / "api" / "config.py", CONFIG_EDIT, "f4 TransferConfig patch")
    sys.modules["skyplane.api.config"] = cfg
    # the import edit applies to the module head, the two plan() edits to the body of MulticastDirectPlanner only (its one-sided
    # subclasses repeat the same lines): patch that class's text, then put the file back together
    ptxt = (REF / "planner" / "planner.py").read_text()
    a, b = ptxt.index("class MulticastDirectPlanner(Planner):"), ptxt.index("class DirectPlannerSourceOneSided(MulticastDirectPlanner):")
    ptxt = _apply(ptxt[:a], PLANNER_EDIT[:1], "f4 planner import") + _apply(ptxt[a:b], PLANNER_EDIT[1:], "f4 planner patch") + ptxt[b:]
    planner = types.ModuleType("skyplane.planner.planner")
    planner.__file__ = str(REF / "planner" / "planner.py")
    exec(compile(ptxt, planner.__file__, "exec"), planner.__dict__)

    class _Iface:
        def __init__(self, tag, bucket):
            self._t, self._b = tag, bucket

        def region_tag(self):
            return self._t

        def bucket(self):
            return self._b

    Path(__import__('os').environ["SKYPLANE_CONFIG"]).write_text("")      # an empty client config: all clouds off, default flags

    def plan_for(tc):
        job = types.SimpleNamespace(src_iface=_Iface("test:src", "srcb"), dst_ifaces=[_Iface("test:dst", "dstb")], dst_prefixes=["p/"], uuid="job0")
        p = planner.MulticastDirectPlanner(n_instances=1, n_connections=32, transfer_config=tc)
        plan = p.plan([job])
        gw = plan.get_region_gateways("test:src")[0]
        return json.loads(plan.get_gateway_program_json(gw.gateway_id)), plan

    def chain(program):
        ops, node = [], program[0]["value"][0]
        while True:
            ops.append(node)
            if not node["children"]:
                return ops
            node = node["children"][0]

    prog_gpu, plan = plan_for(cfg.TransferConfig(use_gpu_compression=True, use_e2ee=False))
    ops = chain(prog_gpu)
    assert [o["op_type"] for o in ops] == ["read_object_store", "gpu_compress", "mux_and", "mux_or", "send"], [o["op_type"] for o in ops]
    assert ops[-1]["compress"] is False and ops[1]["compute_md5"] is True and ops[1]["max_batch"] == 64
    dst_gw = plan.get_region_gateways("test:dst")[0]
    dprog = json.loads(plan.get_gateway_program_json(dst_gw.gateway_id))
    assert dprog[0]["value"][0]["op_type"] == "receive" and dprog[0]["value"][0]["decompress"] is True      # frames are decoded on arrival as before
    # use_gpu_dedup: recipes instead of frames -- gpu_compress(dedup_wire) at the source, gpu_decompress(dedup_wire) in the
    # place of `receive` at the destination, and no effect without use_gpu_compression
    prog_dd, plan_dd = plan_for(cfg.TransferConfig(use_gpu_compression=True, use_gpu_dedup=True, use_e2ee=False))
    ops_dd = chain(prog_dd)
    assert [o["op_type"] for o in ops_dd] == ["read_object_store", "gpu_compress", "mux_and", "mux_or", "send"]
    assert ops_dd[1]["dedup_wire"] is True and ops_dd[1]["cdc"] is True and ops_dd[1]["dedup"] is True and ops_dd[-1]["compress"] is False
    ddst = json.loads(plan_dd.get_gateway_program_json(plan_dd.get_region_gateways("test:dst")[0].gateway_id))
    assert ddst[0]["value"][0]["op_type"] == "gpu_decompress" and ddst[0]["value"][0]["dedup_wire"] is True and ddst[0]["value"][0]["num_workers"] == 1
    assert ddst[0]["value"][0]["children"][0]["op_type"] == "write_object_store"
    assert ops[1]["dedup_wire"] is False
    prog_nd, _ = plan_for(cfg.TransferConfig(use_gpu_dedup=True, use_e2ee=False))
    assert "gpu_compress" not in json.dumps(prog_nd) and "gpu_decompress" not in json.dumps(prog_nd)
    prog_cpu, _ = plan_for(cfg.TransferConfig(use_e2ee=False))
    ops = chain(prog_cpu)
    assert [o["op_type"] for o in ops] == ["read_object_store", "mux_and", "mux_or", "send"] and ops[-1]["compress"] is True   # default unchanged
    prog_off, _ = plan_for(cfg.TransferConfig(use_compression=False, use_gpu_compression=True, use_e2ee=False))
    assert "gpu_compress" not in json.dumps(prog_off)                                                                         # no compression at all
    return [o["op_type"] for o in chain(prog_gpu)]


if __name__ == "__main__":
    n = f3()
    ops = f4()
    print(f"OK f3 chunks={n} f4 program={'>'.join(ops)}")
    sys.stdout.flush()       # (daemon threads of the reference's logger may hold stderr's lock at interpreter shutdown: leave without the teardown, as _reference_dropin.py)
    sys.stderr.flush()
    import os

    os._exit(0)
