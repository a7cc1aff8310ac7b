"""Host-side rendezvous of the ranks of ONE node (one process per GPU).

The path has one exchange, the acquisition peak gather, and that one goes through RCCL (libsgx's `sgx_comm_*`).
What the ranks need besides it is plumbing: a barrier around the timed region, the maximum of a few numbers, and a
way to hand rank 0's RCCL unique id to the others.  That is a star of stream sockets in the abstract AF_UNIX name
space, named after MASTER_PORT: nothing on the file system, nothing left behind when a process dies, and no
dependency beyond the standard library.  Messages are length-prefixed JSON (bytes as base64) - never pickles -, and
rank 0 admits only processes of its own user that present the launcher's token and an unclaimed rank.
"""
import base64
import json
import os
import socket
import struct
import time

MAX_MESSAGE = 1 << 24   # bytes; what crosses is a rank number, a few floats, short strings, the 128-byte RCCL id, peak bytes


def _plain(obj):
    """What may cross the wire: None, bool, int, float, str, bytes and lists / dicts (string keys) of those.  Nothing is
    ever unpickled: a message is JSON, bytes travel as base64 under a reserved key."""
    if obj is None or isinstance(obj, (bool, int, float, str)):
        return obj
    if isinstance(obj, (bytes, bytearray, memoryview)):
        return {"__bytes__": base64.b64encode(bytes(obj)).decode("ascii")}
    if isinstance(obj, (list, tuple)):
        return [_plain(x) for x in obj]
    if isinstance(obj, dict):
        return {str(k): _plain(v) for k, v in obj.items()}
    if hasattr(obj, "item") and getattr(obj, "shape", None) == ():   # numpy scalars
        return _plain(obj.item())
    raise TypeError("rendezvous: %r does not cross the wire" % type(obj))


def _unplain(obj):
    if isinstance(obj, dict):
        if set(obj) == {"__bytes__"}:
            return base64.b64decode(obj["__bytes__"])
        return {k: _unplain(v) for k, v in obj.items()}
    if isinstance(obj, list):
        return [_unplain(x) for x in obj]
    return obj


def _send(sock, obj):
    data = json.dumps(_plain(obj), separators=(",", ":")).encode("utf-8")
    sock.sendall(struct.pack("<Q", len(data)) + data)


def _recv_exact(sock, n):
    buf = bytearray()
    while len(buf) < n:
        part = sock.recv(n - len(buf))
        if not part:
            raise ConnectionError("rendezvous: a rank closed its connection")
        buf += part
    return bytes(buf)


def _recv(sock):
    (n,) = struct.unpack("<Q", _recv_exact(sock, 8))
    if n > MAX_MESSAGE:
        raise ConnectionError("rendezvous: a %d-byte message (limit %d)" % (n, MAX_MESSAGE))
    return _unplain(json.loads(_recv_exact(sock, n).decode("utf-8")))


def _peer_uid(conn):
    """uid of the process at the other end of an AF_UNIX connection (SO_PEERCRED), or None where the platform has none."""
    try:
        cred = conn.getsockopt(socket.SOL_SOCKET, socket.SO_PEERCRED, struct.calcsize("3i"))
        return struct.unpack("3i", cred)[1]
    except (AttributeError, OSError):
        return None


class HostGroup(object):
    """Ranks 1..world-1 connect to rank 0; every collective is a gather to rank 0 followed by a broadcast."""
    name = "host-socket"

    def __init__(self, rank, world, key=None, timeout=600.0):
        self.rank = int(rank)
        self.world = int(world)
        self.timeout = float(timeout)
        key = key if key is not None else os.environ.get("MASTER_PORT", "0")
        self.addr = "\0sgx-rendezvous-%s-%s" % (key, os.environ.get("TORCHELASTIC_RUN_ID", "x"))
        self.peers = {}
        self.sock = None
        if self.world == 1:
            return
        if self.rank == 0:
            srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            srv.bind(self.addr)
            srv.listen(self.world)
            srv.settimeout(self.timeout)
            # The abstract name space has no file permissions: whoever connects must be a process of THIS user
            # (SO_PEERCRED), present the launcher's token when there is one (SGX_RDV_TOKEN, set by bench.py's launch_ranks),
            # and claim a rank in 1 .. world-1 that nobody has claimed before.  Anything else is dropped.
            token = os.environ.get("SGX_RDV_TOKEN", "")
            try:
                while len(self.peers) < self.world - 1:
                    conn, _ = srv.accept()
                    conn.settimeout(self.timeout)
                    try:
                        uid = _peer_uid(conn)
                        hello = _recv(conn)
                        ok = (uid is None or uid == os.getuid()) and isinstance(hello, dict) and \
                            hello.get("token", "") == token and isinstance(hello.get("rank"), int) and \
                            1 <= hello["rank"] < self.world and hello["rank"] not in self.peers
                    except (ConnectionError, ValueError, OSError, TypeError):
                        ok = False
                    if not ok:
                        conn.close()
                        continue
                    self.peers[hello["rank"]] = conn
            finally:
                srv.close()
        else:
            deadline = time.time() + self.timeout
            while True:
                s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
                try:
                    s.connect(self.addr)
                    break
                except (ConnectionRefusedError, FileNotFoundError):
                    s.close()
                    if time.time() > deadline:
                        raise TimeoutError("rendezvous: rank 0 did not appear within %.0f s" % self.timeout)
                    time.sleep(0.02)
            s.settimeout(self.timeout)
            _send(s, {"rank": self.rank, "token": os.environ.get("SGX_RDV_TOKEN", "")})
            self.sock = s

    def get_world_size(self):
        return self.world

    def all_gather_object(self, out, obj):
        """out[r] = rank r's obj on every rank (the signature of torch.distributed.all_gather_object)."""
        if self.world == 1:
            out[0] = obj
            return
        if self.rank == 0:
            got = {0: obj}
            for r, c in self.peers.items():
                got[r] = _recv(c)
            full = [got[r] for r in range(self.world)]
            for c in self.peers.values():
                _send(c, full)
        else:
            _send(self.sock, obj)
            full = _recv(self.sock)
        for r in range(self.world):
            out[r] = full[r]

    def gather(self, obj):
        out = [None] * self.world
        self.all_gather_object(out, obj)
        return out

    def barrier(self):
        self.gather(None)

    def max(self, x):
        return max(self.gather(float(x)))

    def broadcast(self, obj, src=0):
        return self.gather(obj if self.rank == src else None)[src]

    def close(self):
        for c in self.peers.values():
            c.close()
        self.peers = {}
        if self.sock is not None:
            self.sock.close()
            self.sock = None
