"""Program-node class for the new operator, following skyplane/gateway/gateway_program.py:6-97: every attribute
of the node is serialised into the program JSON (:21-25) -- that is how the operator gets its knobs -- and
``create_operator`` is the body of the one ``elif op["op_type"] == "gpu_compress"`` branch a maintainer adds to
GatewayDaemon.create_gateway_operators (skyplane/gateway/gateway_daemon.py:182-268; unknown op_type raises
ValueError at :267-268)."""
import json


class GatewayOperator:
    def __init__(self, op_type):
        self.op_type = op_type
        self.children = []
        self.handle = None

    def add_children(self, children):
        self.children.extend(children)

    def add_child(self, child):
        self.children.append(child)

    def set_handle(self, handle: str):
        self.handle = handle

    def to_dict(self):
        return {**self.__dict__, **{"children": [c.to_dict() for c in self.children]}}

    def to_json(self):
        return json.dumps(self.to_dict())

    def __repr__(self):
        return self.to_json()


class GatewayGpuCompress(GatewayOperator):
    def __init__(self, num_workers: int = 1, max_batch: int = 64, max_chunk_mb: int = 64, compute_md5: bool = True, cdc: bool = False, dedup: bool = False,
                 dedup_wire: bool = False, dedup_epoch_mb: int = 8192, in_slots: int = 0, in_slot_chunk_mb: int = 0):
        super().__init__("gpu_compress")
        self.in_slots = in_slots            # page-locked source slot files per worker the reader downloads into (0 = off, needs the reader-side patch)
        self.in_slot_chunk_mb = in_slot_chunk_mb      # ... of the transfer's chunk size (the planner knows it: multipart_chunk_size_mb); 0 = sized by the first batch
        self.num_workers = num_workers      # one forked worker per GPU is the intended setting
        self.max_batch = max_batch
        self.max_chunk_mb = max_chunk_mb
        self.compute_md5 = compute_md5
        self.cdc = cdc or dedup_wire
        self.dedup = dedup or dedup_wire
        self.dedup_wire = dedup_wire        # recipes instead of frames on the wire (gateway/dedup_wire.py); the destination program must carry gpu_decompress(dedup_wire=True)
        self.dedup_epoch_mb = dedup_epoch_mb


class GatewayGpuDecompress(GatewayOperator):
    """Destination side: takes the place of GatewayReceive's wait operator when the receiver defers the decode."""

    def __init__(self, num_workers: int = 1, max_batch: int = 64, max_chunk_mb: int = 64, verify_md5: bool = True, dedup_wire: bool = False,
                 dedup_store: str = "memory", dedup_verify: str = "segments"):
        super().__init__("gpu_decompress")
        self.dedup_verify = dedup_verify    # how a chunk that travelled as a recipe is checked: "segments" (literal segments against their fingerprints) or "chunk" (its own MD5 chain)
        # the in-memory segment store lives in one worker process (its lanes share it); "files" puts it into the chunk directory for several
        self.num_workers = 1 if (dedup_wire and dedup_store == "memory") else num_workers
        self.dedup_wire = dedup_wire
        self.dedup_store = dedup_store
        self.max_batch = max_batch
        self.max_chunk_mb = max_chunk_mb
        self.verify_md5 = verify_md5


def create_operator(op: dict, handle: str, region: str, input_queue, output_queue, error_event, error_queue, chunk_store):
    """Instantiate the runtime operator from its program-JSON dict (handle = op_type + "_" + handle, daemon :150)."""
    from skyplane_amd.gateway.operators.gateway_operator import GatewayHipCompress, GatewayHipDecompress

    if op["op_type"] == "gpu_decompress":
        return GatewayHipDecompress(handle=handle, region=region, input_queue=input_queue, output_queue=output_queue, error_event=error_event,
                                    error_queue=error_queue, chunk_store=chunk_store, n_processes=op.get("num_workers", 1), max_batch=op.get("max_batch", 64),
                                    max_chunk_bytes=op.get("max_chunk_mb", 64) << 20, verify_md5=op.get("verify_md5", True),
                                    dedup_store=op.get("dedup_store", "memory"), dedup_wire=op.get("dedup_wire", False),
                                    dedup_verify=op.get("dedup_verify", "segments"))      # (recipes are recognised by their magic; dedup_wire only sets the lanes' defaults)
    if op["op_type"] != "gpu_compress":
        raise ValueError(f"Unsupported op_type {op['op_type']}")   # same failure mode as gateway_daemon.py:267-268
    return GatewayHipCompress(handle=handle, region=region, input_queue=input_queue, output_queue=output_queue, error_event=error_event,
                              error_queue=error_queue, chunk_store=chunk_store, n_processes=op.get("num_workers", 1), max_batch=op.get("max_batch", 64),
                              max_chunk_bytes=op.get("max_chunk_mb", 64) << 20, compute_md5=op.get("compute_md5", True), cdc=op.get("cdc", False),
                              dedup=op.get("dedup", False), dedup_wire=op.get("dedup_wire", False), dedup_epoch_bytes=op.get("dedup_epoch_mb", 8192) << 20,
                              in_slots=op.get("in_slots", 0), in_slot_chunk_bytes=op.get("in_slot_chunk_mb", 0) << 20)
