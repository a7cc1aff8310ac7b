"""Host-side rendezvous of the ranks of ONE node (one process per GPU).

The path has one exchange, the acquisition peak gather, and that one goes through RCCL (libsgx's `sgx_comm_*`).
What the ranks need besides it is plumbing: a barrier around the timed region, the maximum of a few numbers, and a
way to hand rank 0's RCCL unique id to the others.  That is a star of stream sockets in the abstract AF_UNIX name
space, named after MASTER_PORT: nothing on the file system, nothing left behind when a process dies, and no
dependency beyond the standard library.
"""
import os
import pickle
import socket
import struct
import time


def _send(sock, obj):
    data = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
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
    return pickle.loads(_recv_exact(sock, n))


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
            try:
                while len(self.peers) < self.world - 1:
                    conn, _ = srv.accept()
                    conn.settimeout(self.timeout)
                    r = _recv(conn)
                    self.peers[int(r)] = conn
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
            _send(s, self.rank)
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
