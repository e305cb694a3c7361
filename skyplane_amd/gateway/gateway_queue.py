"""Operator I/O queues with the interface of the reference's skyplane/gateway/gateway_queue.py
(GatewayQueue :4-28 -- one shared multiprocessing queue; GatewayANDQueue :31-61 -- fan-out to one queue per
registered consumer handle).  Same public method names and semantics, so the gpu_compress operator runs unchanged
against either implementation; inside the reference tree the reference's own classes are used.
"""
from __future__ import annotations

import multiprocessing as mp
from typing import Any, Dict, List, Optional

DEFAULT_DEPTH = 10_000   # the reference bounds every queue at 10000 entries


class GatewayQueue:
    """All registered handles compete for the same stream of chunk requests (work sharing)."""

    def __init__(self, maxsize: int = DEFAULT_DEPTH):
        self._depth = maxsize
        self.q = mp.Queue(maxsize)          # attribute name kept: callers reach into `.q` (e.g. for blocking gets)
        self._consumers: List[str] = []

    # -- wiring -------------------------------------------------------------------------------------
    def register_handle(self, requester_handle: str) -> None:
        self._consumers.append(requester_handle)

    def get_handles(self) -> List[str]:
        return self._consumers

    # -- producers ----------------------------------------------------------------------------------
    def put(self, chunk_req: Any) -> None:
        self.q.put(chunk_req)

    def put_nowait(self, chunk_req: Any) -> None:
        """Raises queue.Full when the bound is reached (the daemon API reports that to the client)."""
        self.q.put_nowait(chunk_req)

    # -- consumers: the handle is accepted for signature compatibility; everyone shares one queue ------
    def get_nowait(self, requester_handle: Optional[str] = None) -> Any:
        return self.q.get_nowait()

    def pop(self, requester_handle: Optional[str] = None) -> None:
        self.q.get()

    def size(self) -> int:
        return self.q.qsize()


class GatewayANDQueue(GatewayQueue):
    """Broadcast: every consumer handle owns a private GatewayQueue and sees every chunk request (mux_and)."""

    def __init__(self, maxsize: int = DEFAULT_DEPTH):
        self._depth = maxsize
        self.q: Dict[str, GatewayQueue] = {}

    def register_handle(self, requester_handle: str) -> None:
        self.q[requester_handle] = GatewayQueue(self._depth)

    def get_handles(self) -> List[str]:
        return [*self.q]

    def get_handle_queue(self, requester_handle: str) -> GatewayQueue:
        return self.q[requester_handle]

    def put(self, chunk_req: Any) -> None:
        for branch in self.q.values():
            branch.put(chunk_req)

    def put_nowait(self, chunk_req: Any) -> None:
        # the reference forbids using a fan-out queue as a pipeline's entry point
        raise ValueError("GatewayANDQueue cannot be the first queue in a pipeline")

    def get_nowait(self, requester_handle: str) -> Any:
        return self.q[requester_handle].get_nowait()

    def pop(self, requester_handle: str) -> None:
        self.q[requester_handle].pop()

    def size(self) -> int:
        return max((branch.size() for branch in self.q.values()), default=0)
