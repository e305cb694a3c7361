"""Mirror of the slice of skyplane/gateway/chunk_store.py the operator touches (ChunkStore :14-109):
chunk files at <chunk_dir>/<chunk_id>.chunk, per-partition request queues, the status-update queue.
``log_chunk_state(..., metadata=...)`` feeds the reference's compression profile hook
(skyplane/gateway/gateway_daemon_api.py:129-134) -- nothing in the reference ever sets that metadata; we do."""
from datetime import datetime
from multiprocessing import Queue
from os import PathLike
from pathlib import Path
from typing import Dict, Optional

from skyplane_amd.chunk import ChunkRequest, ChunkState
from skyplane_amd.gateway import sidecar
from skyplane_amd.gateway.gateway_queue import GatewayQueue
from skyplane_amd.gateway.sidecar import DIGEST_SUFFIX, SIDECAR_SUFFIX   # <id>.chunk.lz4f / <id>.chunk.md5


class ChunkStore:
    def __init__(self, chunk_dir: PathLike):
        self.chunk_dir = Path(chunk_dir)
        self.chunk_dir.mkdir(parents=True, exist_ok=True)
        for f in (list(self.chunk_dir.glob("*.chunk")) + list(self.chunk_dir.glob("*.chunk" + SIDECAR_SUFFIX)) + list(self.chunk_dir.glob("*.chunk" + DIGEST_SUFFIX)) +
                  list(self.chunk_dir.glob("_arena_*.shm"))):      # (arenas of the shared-memory hand-off left by a crashed run)
            f.unlink()
        self.chunk_requests: Dict[str, GatewayQueue] = {}
        self.chunk_status_queue: Queue = Queue()

    def add_partition(self, partition_id: str, queue: GatewayQueue):
        if partition_id in self.chunk_requests:
            raise ValueError(f"Partition {partition_id} already exists")
        self.chunk_requests[partition_id] = queue

    def add_chunk_request(self, chunk_request: ChunkRequest, state: ChunkState = ChunkState.registered):
        pid = chunk_request.chunk.partition_id
        if pid not in self.chunk_requests:
            raise ValueError(f"Partition {pid} does not exist in {self.chunk_requests} - was the gateway program loaded?")
        try:
            self.chunk_requests[pid].put_nowait(chunk_request)
        except Exception:
            return self.chunk_requests[pid].size(), False
        self.log_chunk_state(chunk_request, state)
        return self.chunk_requests[pid].size(), True

    def log_chunk_state(self, chunk_req: ChunkRequest, new_status: ChunkState, worker_id: Optional[int] = None, operator_handle: Optional[str] = None,
                        metadata: Optional[Dict] = None):
        rec = {"chunk_id": chunk_req.chunk.chunk_id, "partition": chunk_req.chunk.partition_id, "state": new_status.name,
               "time": str(datetime.utcnow().isoformat()), "handle": operator_handle, "worker_id": worker_id}
        if metadata is not None:
            rec.update(metadata)
        self.chunk_status_queue.put(rec)

    def get_chunk_file_path(self, chunk_id: str) -> Path:
        return self.chunk_dir / f"{chunk_id}.chunk"

    # -- additions used by the GPU stage and the cooperating sender --------------------------------------
    def get_compressed_file_path(self, chunk_id: str) -> Path:
        return sidecar.compressed_path(self, chunk_id)

    def get_digest_file_path(self, chunk_id: str) -> Path:
        return sidecar.digest_path(self, chunk_id)
