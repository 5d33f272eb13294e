"""Particle sharding rules shared by the host code and documented for the kernels.

The ensemble is partitioned by particle; every rank holds all tables.  There is no data-path exchange of particles:
the only collective is the per-step sum of the tally vector (RCCL all-reduce inside nk_step).
"""
import socket
import struct
import time


def shard_range(n, rank, nranks):
    """Initial particles [lo, hi) owned by `rank` (even split by index)."""
    return (n * rank) // nranks, (n * (rank + 1)) // nranks


def emission_owner(rm, level, step, nranks):
    """Rank that creates the `level`-th particle of reservoir-mode entry `rm` at `step`.  Every rank advances all
    reservoir counters identically and keeps only its own entries (nk_emit_entry / k_emit in csrc/nk_kernels.h)."""
    return (rm + level + step) % nranks


def emission_pid(rm, level, step):
    """64-bit id of an emitted particle: (step+1) mod 2^24 | rm (28 bits) | level (12 bits).  Ids key the counter-based
    RNG, so a particle's random decisions do not depend on which rank or slot holds it."""
    return (((step + 1) & 0xFFFFFF) << 40) | (rm << 12) | level


# --------------------------------------------------------------------------------------------------------------
# One-node rendezvous of the rank processes, standard library only (the engine itself talks RCCL; this is only
# the host-side plumbing a launcher needs: hand the RCCL unique id to every rank, barriers, a max over ranks).
class NodeRendezvous(object):
    """All ranks of ONE node meet on an abstract-namespace UNIX socket that rank 0 listens on.

    `key` names the job (the launcher's MASTER_PORT, or NK_RDV_KEY): ranks of different jobs never meet.  An abstract
    socket exists only while its listener lives, so there are no stale files: a rank that arrives before rank 0 gets
    ECONNREFUSED and retries.  Every operation is an all-gather through rank 0 (payloads of a few bytes)."""

    def __init__(self, rank, world, key, timeout=300.0):
        self.rank, self.world, self.timeout = int(rank), int(world), float(timeout)
        self.addr = '\0nanokappa_rdv_%s' % key
        self.peers, self.sock = [], None
        if self.world <= 1:
            return
        if self.rank == 0:
            srv = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
            srv.bind(self.addr)
            srv.listen(self.world)
            srv.settimeout(self.timeout)
            got = {}
            try:
                while len(got) < self.world - 1:
                    c, _ = srv.accept()
                    c.settimeout(self.timeout)
                    r = struct.unpack('<i', self._recv(c, 4))[0]
                    got[r] = c
            finally:
                srv.close()
            self.peers = [got[r] for r in range(1, self.world)]
        else:
            t0 = time.time()
            while True:
                s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
                try:
                    s.connect(self.addr)
                    break
                except (ConnectionRefusedError, FileNotFoundError):
                    s.close()
                    if time.time() - t0 > self.timeout:
                        raise RuntimeError('rendezvous: rank 0 did not appear on %r' % self.addr)
                    time.sleep(0.05)
            s.settimeout(self.timeout)
            s.sendall(struct.pack('<i', self.rank))
            self.sock = s

    @staticmethod
    def _recv(s, n):
        buf = b''
        while len(buf) < n:
            part = s.recv(n - len(buf))
            if not part:
                raise RuntimeError('rendezvous: a rank went away')
            buf += part
        return buf

    @classmethod
    def _recv_msg(cls, s):
        n = struct.unpack('<q', cls._recv(s, 8))[0]
        return cls._recv(s, n)

    @staticmethod
    def _send_msg(s, b):
        s.sendall(struct.pack('<q', len(b)) + b)

    def allgather(self, payload):
        """Every rank's payload (bytes), in rank order; returns on every rank once all have arrived."""
        if self.world <= 1:
            return [payload]
        if self.rank == 0:
            parts = [payload] + [self._recv_msg(c) for c in self.peers]
            blob = b''.join(struct.pack('<q', len(p)) + p for p in parts)
            for c in self.peers:
                self._send_msg(c, blob)
            return parts
        self._send_msg(self.sock, payload)
        blob = self._recv_msg(self.sock)
        parts, off = [], 0
        for _ in range(self.world):
            n = struct.unpack_from('<q', blob, off)[0]
            parts.append(blob[off + 8:off + 8 + n])
            off += 8 + n
        return parts

    def barrier(self):
        self.allgather(b'')

    def broadcast(self, payload, root=0):
        return self.allgather(payload if self.rank == root else b'')[root]

    def max(self, x):
        return max(struct.unpack('<d', p)[0] for p in self.allgather(struct.pack('<d', float(x))))

    def close(self):
        for c in self.peers:
            c.close()
        if self.sock is not None:
            self.sock.close()
        self.peers, self.sock = [], None
