"""Process launcher and control plane for one process per GPU, standard library only.

The data path of a sharded ensemble is the library's own RCCL communicator (`kbdm_comm_*`, `kbdm_plan_gather`:
device buffers over xGMI).  What a launcher has to add is small: start the ranks, hand rank 0's 128-byte communicator
id to the others, a barrier and a max-over-ranks for the timing.  `Rendezvous` does that over one TCP socket per rank
on 127.0.0.1 (rank 0 listens); `spawn` starts the ranks as child processes.  It also works under an external launcher
that exports RANK / WORLD_SIZE / MASTER_PORT (e.g. `python -m torch.distributed.run`): the port is derived from
MASTER_PORT, nothing of that launcher's own machinery is used.
"""
import hmac
import os
import secrets
import socket
import struct
import subprocess
import sys
import time


def _send(sock, payload):
    sock.sendall(struct.pack("<Q", len(payload)) + payload)


def _recv(sock):
    hdr = b""
    while len(hdr) < 8:
        part = sock.recv(8 - len(hdr))
        if not part:
            raise ConnectionError("rendezvous peer closed the connection")
        hdr += part
    n, = struct.unpack("<Q", hdr)
    buf = bytearray()
    while len(buf) < n:
        part = sock.recv(min(1 << 20, n - len(buf)))
        if not part:
            raise ConnectionError("rendezvous peer closed the connection")
        buf += part
    return bytes(buf)


def _pack_parts(parts):
    """[bytes] -> count + (length, bytes) pairs: the payloads are opaque byte strings, nothing is ever unpickled."""
    return struct.pack("<I", len(parts)) + b"".join(struct.pack("<Q", len(b)) + b for b in parts)


def _unpack_parts(blob, world):
    n, = struct.unpack_from("<I", blob, 0)
    if n != world:
        raise ConnectionError(f"rendezvous: {n} parts for a world of {world}")
    out, o = [], 4
    for _ in range(n):
        ln, = struct.unpack_from("<Q", blob, o)
        o += 8
        if o + ln > len(blob):
            raise ConnectionError("rendezvous: truncated frame")
        out.append(blob[o:o + ln])
        o += ln
    return out


def rendezvous_port():
    """KBDM_RDZV_PORT, or a port next to an external launcher's MASTER_PORT (that port itself belongs to the launcher)."""
    if "KBDM_RDZV_PORT" in os.environ:
        return int(os.environ["KBDM_RDZV_PORT"])
    base = int(os.environ.get("MASTER_PORT", "29400"))
    return 20000 + (base * 7 + 4111) % 30000


class Rendezvous:
    """Star topology on rank 0: `allgather(bytes) -> [bytes] * world` is the one primitive; `bcast`, `barrier` and
    `max` are built on it.  Every rank must make the same sequence of calls."""

    def __init__(self, rank=None, world=None, port=None, timeout=300.0):
        self.rank = int(os.environ.get("RANK", "0")) if rank is None else int(rank)
        self.world = int(os.environ.get("WORLD_SIZE", "1")) if world is None else int(world)
        self.port = rendezvous_port() if port is None else int(port)
        self.timeout = float(timeout)
        # `spawn` hands every rank a random token through the environment: a local process that does not know it
        # cannot pose as a rank (the hello of a connecting rank is rank + token)
        self.token = os.environ.get("KBDM_RDZV_TOKEN", "").encode()
        self.peers, self.sock, self.server = [], None, None
        if self.world == 1:
            return
        if self.rank == 0:
            self.server = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            self.server.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            self.server.bind(("127.0.0.1", self.port))
            self.server.listen(self.world)
            self.server.settimeout(timeout)
            peers = {}
            while len(peers) < self.world - 1:
                conn, _ = self.server.accept()
                conn.settimeout(timeout)            # a peer that dies later raises here instead of hanging rank 0
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                try:
                    hello = _recv(conn)
                    r, = struct.unpack_from("<I", hello, 0)
                    ok = 1 <= r < self.world and r not in peers and hmac.compare_digest(hello[4:], self.token)
                except (ConnectionError, OSError, struct.error):
                    ok = False
                if not ok:                          # not one of ours (bad rank, duplicate, wrong token): drop it
                    conn.close()
                    continue
                peers[r] = conn
            self.peers = [peers[r] for r in range(1, self.world)]
        else:
            t0 = time.time()
            while True:
                try:
                    self.sock = socket.create_connection(("127.0.0.1", self.port), timeout=timeout)
                    break
                except OSError:
                    if time.time() - t0 > timeout:
                        raise
                    time.sleep(0.05)
            self.sock.settimeout(timeout)
            self.sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            _send(self.sock, struct.pack("<I", self.rank) + self.token)

    def allgather(self, payload):
        payload = bytes(payload)
        if self.world == 1:
            return [payload]
        if self.rank == 0:
            parts = [payload] + [_recv(c) for c in self.peers]
            blob = _pack_parts(parts)
            for c in self.peers:
                _send(c, blob)
            return parts
        _send(self.sock, payload)
        return _unpack_parts(_recv(self.sock), self.world)

    def bcast(self, payload, src=0):
        return self.allgather(payload if self.rank == src else b"")[src]

    def barrier(self):
        self.allgather(b"")

    def max(self, value):
        return max(struct.unpack("<d", b)[0] for b in self.allgather(struct.pack("<d", float(value))))

    def exchange_id(self, uid):
        """The `exchange_id` callable of `distributed.RcclComm`: rank 0 passes the id it created, everyone gets it."""
        return self.bcast(uid or b"", src=0)

    def close(self):
        for c in self.peers:
            c.close()
        if self.sock:
            self.sock.close()
        if self.server:
            self.server.close()
        self.peers, self.sock, self.server = [], None, None


def spawn(argv, world, env=None, port=None, poll=0.05, grace=5.0):
    """Start `world` ranks of `argv` (a command line) as child processes with RANK / LOCAL_RANK / WORLD_SIZE /
    KBDM_RDZV_PORT / KBDM_RDZV_TOKEN set and watch ALL of them: when one exits with a non-zero code (LinAlgError, a HIP
    error, out of memory) the others - which may sit in the gather or in a rendezvous receive waiting for it - are
    terminated, then killed after `grace` seconds, and that code is returned (0 if all succeeded).  Children only: the
    calling process is never replaced."""
    if port is None:
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
    token = secrets.token_hex(16)
    procs = []
    for r in range(world):
        e = dict(os.environ if env is None else env)
        e.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), KBDM_RDZV_PORT=str(port), KBDM_RDZV_TOKEN=token)
        procs.append(subprocess.Popen(list(argv), env=e))
    rc = 0
    try:
        live = list(procs)
        while live and rc == 0:
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0:
                    rc = code
                    break
            if live and rc == 0:
                time.sleep(poll)
    finally:
        live = [p for p in procs if p.poll() is None]
        for p in live:                              # exactly the processes started above, by handle
            p.terminate()
        t0 = time.time()
        for p in live:
            try:
                p.wait(timeout=max(0.0, grace - (time.time() - t0)))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
    return rc


if __name__ == "__main__":      # python -m llckbdm_amd.launch N script.py args...
    sys.exit(spawn([sys.executable] + sys.argv[2:], int(sys.argv[1])))
