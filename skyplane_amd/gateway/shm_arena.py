"""Shared-memory hand-off of wire payloads between the GPU operators and the processes on either side of them (SURVEY.md 8f item 2:
"pinned shared memory instead of tmpfs files").

The reference hands a chunk from operator to operator as ONE FILE PER CHUNK in the gateway's chunk directory (a tmpfs;
``ChunkStore.get_chunk_file_path``, skyplane/gateway/chunk_store.py:108-109): GatewaySender re-reads it with ``f.read()``
(gateway_operator.py:350-351), GatewayReceiver writes it with ``f.write()`` (gateway_receiver.py:204-211).  With gpu_compress upstream that
meant: frame -> pinned staging -> ``write()`` into ``<id>.chunk.lz4f`` -> ``read()``/sendfile.  Here the frame's home is a slot of an ARENA: one
file in the same directory, mapped MAP_SHARED by everybody who needs it and page-locked once by the process that owns a device context
(``skyhip_host_register``), so that

    source:       device --DMA--> arena slot --sendfile--> socket                        (no CPU copy of the frame)
    destination:  socket --recv_into--> arena slot --DMA--> device                       (one copy: kernel socket buffer -> arena)

What the reference's ChunkStore contract still sees is a file with the sidecar's name (``<id>.chunk.lz4f``): a 26-byte-plus-name POINTER
(magic "SKYSHM1\\0", offset, length, arena file name) written under a temporary name and renamed, so its existence still means "the payload is
complete", and unlinking it -- what the consumer does when it is done with the payload -- is what frees the slot.  A payload that starts with
the LZ4 frame magic or a recipe's "SKYD" is a plain payload file as before: both forms are accepted everywhere (``open_payload``), and a
producer that finds no free slot simply falls back to writing the file.  Nothing here touches the wire format or chunk.py."""
from __future__ import annotations

import mmap
import os
import socket
import time
import struct
import threading
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import numpy as np

MAGIC = b"SKYSHM1\x00"
_PTR = struct.Struct("<8sQQH")
_HDR = struct.Struct("<8sQQ")          # arena file header: magic, slot_bytes, n_slots (first page)
HEADER_BYTES = 4096
ARENA_PREFIX = "_arena_"
ARENA_SUFFIX = ".shm"


class ArenaError(ValueError):
    """A pointer file or arena that is not what it claims to be."""


class ShmArena:
    """``n_slots`` slots of ``slot_bytes`` in one file, mapped shared.  create=True makes (and sizes) the file; otherwise it is opened."""

    def __init__(self, path, slot_bytes: int = 0, n_slots: int = 0, create: bool = False):
        self.path = Path(path)
        if create:
            slot_bytes = (int(slot_bytes) + 4095) & ~4095
            assert slot_bytes > 0 and n_slots > 0
            fd = os.open(self.path, os.O_RDWR | os.O_CREAT | os.O_EXCL, 0o600)
            os.ftruncate(fd, HEADER_BYTES + slot_bytes * n_slots)
            os.pwrite(fd, _HDR.pack(MAGIC, slot_bytes, n_slots), 0)
        else:
            fd = os.open(self.path, os.O_RDWR)
            magic, slot_bytes, n_slots = _HDR.unpack(os.pread(fd, _HDR.size, 0))
            if magic != MAGIC or slot_bytes <= 0 or n_slots <= 0 or os.fstat(fd).st_size < HEADER_BYTES + slot_bytes * n_slots:
                os.close(fd)
                raise ArenaError(f"{self.path} is not an arena file")
        self.fd, self.slot_bytes, self.n_slots = fd, int(slot_bytes), int(n_slots)
        self.size = HEADER_BYTES + self.slot_bytes * self.n_slots
        self._mm = mmap.mmap(fd, self.size, mmap.MAP_SHARED, mmap.PROT_READ | mmap.PROT_WRITE)
        self.buf = np.frombuffer(self._mm, np.uint8)         # the whole mapping (what skyhip_host_register page-locks)

    def slot_offset(self, i: int) -> int:
        return HEADER_BYTES + i * self.slot_bytes

    def slot(self, i: int) -> np.ndarray:
        o = self.slot_offset(i)
        return self.buf[o:o + self.slot_bytes]

    def view(self, off: int, length: int) -> np.ndarray:
        if off < HEADER_BYTES or off + length > self.size:
            raise ArenaError(f"range {off}+{length} lies outside arena {self.path.name}")
        return self.buf[off:off + length]

    def close(self, unlink: bool = False):
        if unlink:
            try:
                self.path.unlink()
            except FileNotFoundError:
                pass
        try:
            self.buf = None
            self._mm.close()
        except (BufferError, ValueError):
            pass                                               # views still alive somewhere: the mapping goes with them
        try:
            os.close(self.fd)
        except OSError:
            pass


def write_pointer(pointer_path: Path, arena: ShmArena, off: int, length: int):
    """<id>.chunk.lz4f := pointer to arena[off : off + length], visible only when complete (temporary name + rename)."""
    name = arena.path.name.encode()
    tmp = pointer_path.with_name(pointer_path.name + ".ptmp")
    with open(tmp, "wb") as f:
        f.write(_PTR.pack(MAGIC, off, length, len(name)) + name)
    os.replace(tmp, pointer_path)


def parse_pointer(blob: bytes) -> Optional[Tuple[str, int, int]]:
    """(arena file name, offset, length) when blob is a pointer; None when it is a plain payload."""
    if len(blob) < 8 or blob[:8] != MAGIC:
        return None
    if len(blob) < _PTR.size:
        raise ArenaError("truncated pointer file")
    _, off, length, nlen = _PTR.unpack_from(blob, 0)
    name = blob[_PTR.size:_PTR.size + nlen].decode("utf-8", "replace")
    if len(blob) != _PTR.size + nlen or not name.startswith(ARENA_PREFIX) or not name.endswith(ARENA_SUFFIX) or "/" in name or "\x00" in name:
        raise ArenaError(f"malformed pointer file (arena name {name!r})")
    return name, int(off), int(length)


class ArenaWriter:
    """Producer side: hands out free slots of ONE arena and publishes pointers to them.  A slot is free again when the pointer that was published
    for it no longer exists (the consumer unlinked it) -- a stat() per candidate slot, no other IPC.  Thread-safe."""

    def __init__(self, directory, tag: str, slot_bytes: int, n_slots: int):
        self.dir = Path(directory)
        self.arena = ShmArena(self.dir / f"{ARENA_PREFIX}{tag}{ARENA_SUFFIX}", slot_bytes, n_slots, create=True)
        self._owner: List[Optional[Path]] = [None] * n_slots
        self._busy = [False] * n_slots                        # taken, not yet published
        self._next = 0
        self._lock = threading.Lock()

    def take(self, want: int) -> List[int]:
        """Up to `want` free slots (possibly none: the caller then writes plain files)."""
        got: List[int] = []
        with self._lock:
            n = self.arena.n_slots
            for k in range(n):
                if len(got) >= want:
                    break
                i = (self._next + k) % n
                if self._busy[i]:
                    continue
                own = self._owner[i]
                if own is not None and own.exists():
                    continue
                self._owner[i] = None
                self._busy[i] = True
                got.append(i)
            if got:
                self._next = (got[-1] + 1) % n
        return got

    def publish(self, slot: int, pointer_path: Path, length: int):
        assert 0 <= length <= self.arena.slot_bytes
        write_pointer(pointer_path, self.arena, self.arena.slot_offset(slot), length)
        with self._lock:
            self._owner[slot] = pointer_path
            self._busy[slot] = False

    def give_back(self, slot: int):
        with self._lock:
            self._busy[slot] = False

    def close(self):
        self.arena.close(unlink=True)


# ---- consumer side: a per-process cache of opened arenas ----
_open_lock = threading.Lock()
_opened: Dict[str, ShmArena] = {}
_registered: Dict[str, object] = {}


def _arena_for(pointer_path: Path, name: str) -> ShmArena:
    key = str(pointer_path.parent / name)
    with _open_lock:
        a = _opened.get(key)
        if a is None:
            a = ShmArena(key)
            _opened[key] = a
        return a


def register_once(arena: ShmArena, ctx) -> bool:
    """Page-lock the arena for this PROCESS through ctx (hipHostRegister is per process: the lanes of a worker share one registration).
    Returns True when this call did it.  A context without register_host (emulator, null device) needs none."""
    if not hasattr(ctx, "register_host"):
        return False
    key = str(arena.path)
    with _open_lock:
        if key in _registered:
            return False
        ctx.register_host(arena.buf)
        _registered[key] = ctx
        return True


def forget(arena: ShmArena, ctx=None):
    key = str(arena.path)
    with _open_lock:
        owner = _registered.pop(key, None)
        _opened.pop(key, None)
    if owner is not None and hasattr(owner, "unregister_host"):
        try:
            owner.unregister_host(arena.buf)
        except Exception:
            pass


class Payload:
    """What a sidecar path leads to: `view` (uint8 array over the arena, zero-copy) for a pointer, otherwise a plain file to be read."""

    __slots__ = ("path", "arena", "off", "length")

    def __init__(self, path: Path, arena: Optional[ShmArena], off: int, length: int):
        self.path, self.arena, self.off, self.length = path, arena, off, length

    @property
    def view(self) -> Optional[np.ndarray]:
        return None if self.arena is None else self.arena.view(self.off, self.length)


def open_payload(path) -> Payload:
    """Resolve `<id>.chunk.lz4f`: a pointer into an arena of the same directory, or the payload itself."""
    path = Path(path)
    with open(path, "rb") as f:
        head = f.read(8)
        if head != MAGIC:
            return Payload(path, None, 0, os.fstat(f.fileno()).st_size)
        blob = head + f.read(4096)
    name, off, length = parse_pointer(blob)
    arena = _arena_for(path, name)
    arena.view(off, length)                                   # bounds check now, not at first use
    return Payload(path, arena, off, length)


def read_payload(path) -> bytes:
    """The payload's bytes whichever form it has (what a sender that wants ``data = f.read()`` calls: INTEGRATION.md section 6)."""
    p = open_payload(path)
    if p.arena is None:
        return Path(path).read_bytes()
    return p.view.tobytes()


def take_payload(path, blob: bytes) -> bytes:
    """For a sender that has just done ``data = f.read()`` on the sidecar (the reference's GatewaySender after INTEGRATION.md section 6): `blob`
    itself when the sidecar is the payload; the slot's bytes when it is a pointer -- the pointer is unlinked right away (the sender holds the bytes
    now, as it does in the reference), which gives the slot back to gpu_compress.  Should the sender fail and the chunk be re-queued, it finds no
    sidecar and takes the reference's own path (raw chunk, CPU compression): slower, never wrong."""
    ptr = parse_pointer(blob)
    if ptr is None:
        return blob
    path = Path(path)
    arena = _arena_for(path, ptr[0])
    data = arena.view(ptr[1], ptr[2]).tobytes()
    try:
        path.unlink()
    except FileNotFoundError:
        pass
    return data


def sendfile_payload(sock, path) -> int:
    """Payload straight to a socket: arena (or file) pages -> socket buffer, no pass through Python bytes.  Returns the bytes sent."""
    p = open_payload(path)
    if p.arena is None:
        with open(path, "rb") as f:
            return sock.sendfile(f, 0, p.length) if p.length else 0
    sent, fd_out = 0, sock.fileno()
    if hasattr(sock, "_sslobj"):                              # TLS: no kernel path, plain sends of the view
        sock.sendall(p.view)
        return p.length
    tmo = sock.gettimeout()
    deadline = None if tmo is None else time.monotonic() + tmo      # the socket's timeout bounds every wait for buffer space, like sock.sendall
    while sent < p.length:
        try:
            n = os.sendfile(fd_out, p.arena.fd, p.off + sent, p.length - sent)
        except BlockingIOError:
            if tmo == 0:
                raise
            import select

            left = None if deadline is None else deadline - time.monotonic()
            if left is not None and left <= 0 or not select.select([], [fd_out], [], left)[1]:
                raise socket.timeout(f"timed out after {sent} of {p.length} payload bytes")       # a stalled peer: do not spin for ever
            continue
        if deadline is not None:
            deadline = time.monotonic() + tmo
        if n == 0:
            raise ConnectionError(f"socket closed after {sent} of {p.length} payload bytes")
        sent += n
    return sent


# ---- destination, raw side: the decoded chunk's file IS page-locked memory (round 5) ----
class LinkSlots:
    """`n_slots` files of exactly `size` bytes in the chunk directory, each mapped MAP_SHARED (and page-locked once by the process that owns the device
    context: `register`).  gpu_decompress lets the device write a decoded chunk of `size` bytes straight into a free slot's pages and PUBLISHES it by
    hard-linking the slot file to ``<id>.chunk`` (temporary name + rename: complete when visible) -- what the reference's neighbours
    (GatewayWaitReceiver's size test, write_object_store's ``upload_object(src_file_path=...)``, gateway_operator.py:125-149, :625-645) then see is an
    ordinary file of the right length whose bytes never passed through a ``write()``.  The daemon deleting ``<id>.chunk`` when the chunk is done drops
    the link count of the slot's inode back to 1, which is what marks the slot free: no other IPC.  Chunks of another length (an object's short tail)
    and times when every slot is still waiting for its upload take the plain write path -- slower, never wrong.

    The source side has its own slot files since round 6 (InSlots below) -- behind a two-word change in the reference, because there ``<id>.chunk`` is created by ``download_object(..., dst_file_path)`` of whichever object-store interface
    serves the bucket (s3_interface.py:156-192 and siblings), every one of which opens its destination with mode "wb" -- a truncation, which frees the
    pages a registration pinned (INTEGRATION 6e is the change)."""

    def __init__(self, directory, tag: str, size: int, n_slots: int):
        self.dir, self.size, self.n = Path(directory), int(size), int(n_slots)
        assert self.size > 0 and self.n > 0
        self.paths: List[Path] = []
        self._maps: List[mmap.mmap] = []
        self.views: List[np.ndarray] = []
        self._busy = [False] * self.n
        self._next = 0
        self._lock = threading.Lock()
        self._registered_by = None
        try:
            for k in range(self.n):
                p = self.dir / f"_outslot_{tag}_{k}.bin"
                fd = os.open(p, os.O_RDWR | os.O_CREAT | os.O_EXCL, 0o644)
                self.paths.append(p)                   # (from here on close() removes it)
                try:
                    # the blocks are ALLOCATED now: a sparse file on a full tmpfs raises SIGBUS at the first touch through the mapping --
                    # the process dies -- where posix_fallocate raises ENOSPC, which the operator answers with the plain write path
                    os.posix_fallocate(fd, 0, self.size)
                    mm = mmap.mmap(fd, self.size, mmap.MAP_SHARED, mmap.PROT_READ | mmap.PROT_WRITE)
                finally:
                    os.close(fd)
                self._maps.append(mm)
                self.views.append(np.frombuffer(mm, np.uint8))
        except BaseException:
            self.close()                               # no slot file is left behind by a half-made set
            raise

    def register(self, ctx) -> bool:
        """Page-lock every slot through ctx (once; a context without register_host -- emulator, null device -- needs none)."""
        if self._registered_by is not None or not hasattr(ctx, "register_host"):
            return False
        self._registered_by = ctx              # (set first: close() after a failure half way unregisters what was registered; unregistering the rest fails quietly)
        for v in self.views:
            v[::4096] = 0                      # touch: the pages exist (posix_fallocate made sure they can) before they are pinned
            ctx.register_host(v)
        return True

    def take(self, want: int) -> List[int]:
        got: List[int] = []
        with self._lock:
            for k in range(self.n):
                if len(got) >= want:
                    break
                i = (self._next + k) % self.n
                if self._busy[i]:
                    continue
                try:
                    if os.stat(self.paths[i]).st_nlink != 1:      # still published: its chunk has not been uploaded and deleted yet
                        continue
                except FileNotFoundError:
                    continue
                self._busy[i] = True
                got.append(i)
            if got:
                self._next = (got[-1] + 1) % self.n
        return got

    def idle(self) -> bool:
        """No slot is taken or still published (every chunk that went through one has been uploaded and deleted)."""
        with self._lock:
            if any(self._busy):
                return False
            try:
                return all(os.stat(p).st_nlink == 1 for p in self.paths)
            except FileNotFoundError:
                return False

    def publish(self, slot: int, final: Path):
        tmp = final.with_name(final.name + ".lnk")
        try:
            os.unlink(tmp)
        except FileNotFoundError:
            pass
        os.link(self.paths[slot], tmp)
        try:
            os.replace(tmp, final)
        except BaseException:
            try:
                os.unlink(tmp)                 # (a .lnk left behind keeps the slot's link count at 2: the slot would never be free again)
            except OSError:
                pass
            raise
        finally:
            with self._lock:
                self._busy[slot] = False

    def give_back(self, slot: int):
        with self._lock:
            self._busy[slot] = False

    def close(self):
        ctx, self._registered_by = self._registered_by, None
        for v in self.views:
            if ctx is not None and hasattr(ctx, "unregister_host"):
                try:
                    ctx.unregister_host(v)
                except Exception:
                    pass
        for p in self.paths:
            try:
                p.unlink()                     # a published link keeps the inode (and its bytes) alive for whoever still reads it
            except FileNotFoundError:
                pass
        self.paths = []
        self.views = []
        for mm in self._maps:
            try:
                mm.close()
            except (BufferError, ValueError):
                pass
        self._maps = []


# ---- source, raw side: the READER's file write lands in page-locked memory (round 6, SURVEY 8f item 2 "and the reader's file write") ----
IN_PREFIX = "_inslot_"


class InSlots:
    """Files of exactly `size` bytes in the SOURCE gateway's chunk directory, made, mapped MAP_SHARED and page-locked by gpu_compress's worker -- the mirror
    image of LinkSlots.  Here the operator is the CONSUMER: whoever writes ``<id>.chunk`` (the reference's GatewayObjStoreReadOperator through
    ``download_object``, gateway_operator.py:555-575) first makes that name a hard link to a free slot (``claim_slot`` below: one call in the reader,
    INTEGRATION.md 6e) and then writes the chunk's bytes into the EXISTING file -- which needs the object-store interfaces to open an existing destination
    without truncating it (``"r+b"`` instead of ``"wb"``: s3_interface.py:183, posix_file_interface.py:105,109 and their siblings; a truncation would free
    the pinned pages).  gpu_compress then finds that ``<id>.chunk``'s inode is one of its slots and hands the device the mapped, page-locked bytes: the
    page cache -> staging copy of `_read_chunks` (one read of every chunk by the CPU) is gone.  The daemon deleting ``<id>.chunk`` when the chunk has
    been sent frees the slot (link count back to 1), as on the destination.  A chunk of another length, a reader without the patch, or no free slot: the
    file is an ordinary file and is read as before."""

    def __init__(self, directory, tag: str, size: int, n_slots: int):
        self.dir, self.size, self.n = Path(directory), int(size), int(n_slots)
        assert self.size > 0 and self.n > 0
        self.paths: List[Path] = []
        self._maps: List[mmap.mmap] = []
        self.views: List[np.ndarray] = []
        self.by_inode: Dict[int, int] = {}
        self._registered_by = None
        try:
            for k in range(self.n):
                p = self.dir / f"{IN_PREFIX}{tag}_{k}_{self.size}.bin.tmp"      # (invisible to claim_slot until it is complete AND page-locked: register renames it)
                fd = os.open(p, os.O_RDWR | os.O_CREAT | os.O_EXCL, 0o644)
                self.paths.append(p)
                try:
                    os.posix_fallocate(fd, 0, self.size)
                    mm = mmap.mmap(fd, self.size, mmap.MAP_SHARED, mmap.PROT_READ | mmap.PROT_WRITE)
                    self.by_inode[os.fstat(fd).st_ino] = k
                finally:
                    os.close(fd)
                self._maps.append(mm)
                self.views.append(np.frombuffer(mm, np.uint8))
        except BaseException:
            self.close()
            raise

    def register(self, ctx):
        """Page-lock every slot through ctx (a context without register_host -- emulator, null device -- needs none), then make the slots visible."""
        self._registered_by = ctx if hasattr(ctx, "register_host") else None
        for v in self.views:
            v[::4096] = 0
            if self._registered_by is not None:
                ctx.register_host(v)
        for k, p in enumerate(self.paths):
            final = p.with_name(p.name[:-4])
            os.replace(p, final)
            self.paths[k] = final

    def view_of(self, st: os.stat_result, nbytes: int) -> Optional[np.ndarray]:
        """The page-locked bytes behind a chunk file whose stat() is `st`, or None when the file is not (a link to) one of these slots."""
        k = self.by_inode.get(st.st_ino)
        if k is None or st.st_size != self.size or nbytes != self.size or st.st_dev != os.stat(self.paths[k]).st_dev:
            return None
        return self.views[k]

    def close(self):
        ctx, self._registered_by = self._registered_by, None
        for v in self.views:
            if ctx is not None and hasattr(ctx, "unregister_host"):
                try:
                    ctx.unregister_host(v)
                except Exception:
                    pass
        for p in self.paths:
            try:
                p.unlink()                     # (a chunk that is still linked keeps the inode and its bytes)
            except FileNotFoundError:
                pass
        self.paths, self.views, self.by_inode = [], [], {}
        for mm in self._maps:
            try:
                mm.close()
            except (BufferError, ValueError):
                pass
        self._maps = []


_claim_cache: Dict[Tuple[str, int], Tuple[float, List[str]]] = {}
_claim_cursor: Dict[Tuple[str, int], int] = {}


def claim_slot(chunk_path, nbytes: int) -> bool:
    """Reader side (any process, no state): make `chunk_path` a hard link to a free source slot of exactly `nbytes` bytes; False = there is none (write
    the file as before).  Lock-free: link, then look at the slot's link count -- 2 means the slot is mine; more means another reader linked it at the
    same moment: let go and try the next one (whoever saw 2 keeps it; both may let go, and the slot is simply free again).  The caller then writes the
    chunk with a NON-truncating open."""
    chunk_path = Path(chunk_path)
    d = chunk_path.parent
    key = (str(d), int(nbytes))
    now = time.monotonic()
    hit = _claim_cache.get(key)
    if hit is None or now - hit[0] > 1.0:
        suffix = f"_{int(nbytes)}.bin"
        try:
            names = sorted(n for n in os.listdir(d) if n.startswith(IN_PREFIX) and n.endswith(suffix))
        except FileNotFoundError:
            names = []
        hit = _claim_cache[key] = (now, names)
    names = hit[1]
    if not names:
        return False
    # Slots come free in roughly the order they were taken (chunks are sent in the order they were read), so the search goes on from where this process's last
    # claim ended: the slot behind it is the one that has been busy longest.  (A random start cost ~n/2 stat() calls per claim once most slots were busy --
    # with 256 slots that was a third of a reader's time, GPU call r6f.)
    for _sweep in range(3):                    # (a slot two readers let go of at the same moment is free again: only a sweep that saw a collision is repeated)
        collided = False
        start = _claim_cursor.get(key, os.getpid() * 7919) % len(names)
        for j in range(len(names)):
            _claim_cursor[key] = (start + j + 1) % len(names)
            p = d / names[(start + j) % len(names)]
            try:
                if os.stat(p).st_nlink != 1:
                    continue
                os.link(p, chunk_path)
            except FileExistsError:
                return False                   # the chunk's file is there already (a retry): whatever it is, it is written in place
            except OSError:
                continue
            try:
                if os.stat(p).st_nlink == 2:
                    return True
            except OSError:
                pass
            collided = True
            try:
                os.unlink(chunk_path)
            except OSError:
                pass
        if not collided:
            break
    return False
