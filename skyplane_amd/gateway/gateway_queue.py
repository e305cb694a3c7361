"""Mirror of skyplane/gateway/gateway_queue.py (GatewayQueue :4-28, GatewayANDQueue :31-61): the operator I/O
type.  Same method names and semantics; inside the reference tree the operator uses the reference's classes."""
from multiprocessing import Queue


class GatewayQueue:
    def __init__(self, maxsize=10000):
        self.q = Queue(maxsize)
        self.handles = []

    def register_handle(self, requester_handle):
        self.handles.append(requester_handle)

    def put(self, chunk_req):
        self.q.put(chunk_req)

    def put_nowait(self, chunk_req):
        self.q.put_nowait(chunk_req)

    def pop(self, requester_handle=None):
        self.q.get()

    def get_nowait(self, requester_handle=None):
        return self.q.get_nowait()

    def get_handles(self):
        return self.handles

    def size(self):
        return self.q.qsize()


class GatewayANDQueue(GatewayQueue):
    """Fan-out: a chunk put here lands in every registered handle's queue."""

    def __init__(self, maxsize=10000):
        self.q = {}
        self.maxsize = maxsize

    def register_handle(self, requester_handle):
        self.q[requester_handle] = GatewayQueue(self.maxsize)

    def get_handles(self):
        return list(self.q.keys())

    def get_handle_queue(self, requester_handle):
        return self.q[requester_handle]

    def put(self, chunk_req):
        for handle in self.q:
            self.q[handle].put(chunk_req)

    def put_nowait(self, chunk_req):
        raise ValueError("GatewayANDQueue cannot be the first queue in a pipeline")

    def pop(self, requester_handle):
        self.q[requester_handle].get()

    def get_nowait(self, requester_handle):
        return self.q[requester_handle].get_nowait()

    def size(self):
        return max((q.size() for q in self.q.values()), default=0)
