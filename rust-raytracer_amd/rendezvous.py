"""Host-side rendezvous of the ranks of one node for bench.py and the multi-GPU binding: a plain TCP star on the
loopback interface.  It carries the 128-byte RCCL id, the barriers around the timed region and two small
reductions -- nothing the GPU sees.  No torch: a process that imports torch binds torch's bundled librccl / HIP
runtime first, and the library would then run its gather on a runtime that none of its tests ran on.

Rank 0 listens on an ephemeral port and publishes it in a file that the launcher's children agree on
(RTX_RDZV_FILE, else a name made from MASTER_ADDR, MASTER_PORT and the launcher's pid: every rank that
torch.distributed.run starts on a node is a child of the same agent process).  MASTER_PORT itself belongs to
the launcher's own store and is left alone."""
import json
import os
import socket
import struct
import tempfile
import time


def _send(sock, obj):
    data = json.dumps(obj).encode()
    sock.sendall(struct.pack("<I", len(data)) + data)


def _recv(sock):
    def rd(n):
        buf = b""
        while len(buf) < n:
            part = sock.recv(n - len(buf))
            if not part:
                raise ConnectionError("rendezvous peer closed the connection")
            buf += part
        return buf
    (n,) = struct.unpack("<I", rd(4))
    return json.loads(rd(n).decode())


class Rendezvous:
    def __init__(self, rank, world, timeout=120.0, path=None):
        self.rank, self.world = rank, world
        self.peers, self.sock, self.path = [], None, None
        if world == 1:
            return
        host = os.environ.get("MASTER_ADDR", "127.0.0.1")
        path = path or os.environ.get("RTX_RDZV_FILE") or os.path.join(
            tempfile.gettempdir(), f"rtx_rdzv_{host}_{os.environ.get('MASTER_PORT', '0')}_{os.getppid()}")
        deadline = time.time() + timeout
        if rank == 0:
            srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
            srv.bind(("127.0.0.1" if host in ("127.0.0.1", "localhost") else host, 0))
            srv.listen(world)
            tmp = f"{path}.{os.getpid()}"
            with open(tmp, "w") as f:
                json.dump({"port": srv.getsockname()[1], "pid": os.getpid()}, f)
            os.replace(tmp, path)          # atomic: a reader sees the whole file or none
            self.path = path
            srv.settimeout(timeout)
            peers = {}
            while len(peers) < world - 1:
                conn, _ = srv.accept()
                conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                peers[_recv(conn)["rank"]] = conn
            srv.close()
            self.peers = [peers[r] for r in range(1, world)]
        else:
            addr = "127.0.0.1" if host in ("127.0.0.1", "localhost") else host
            while self.sock is None:
                # the file is read again after every failed attempt: one left behind by an earlier job of the same
                # name (a launcher pid used twice) names a port nobody listens on until rank 0 has replaced it
                try:
                    with open(path) as f:
                        port = json.load(f)["port"]
                    self.sock = socket.create_connection((addr, port), timeout=timeout)
                except (OSError, ValueError, KeyError):
                    self.sock = None
                    if time.time() > deadline:
                        raise TimeoutError(f"no rendezvous with rank 0 through {path}")
                    time.sleep(0.02)
            self.sock.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            _send(self.sock, {"rank": rank})

    def allgather(self, obj):
        """Every rank's `obj` (JSON-serialisable), in rank order, on every rank."""
        if self.world == 1:
            return [obj]
        if self.rank == 0:
            out = [obj] + [_recv(p) for p in self.peers]
            for p in self.peers:
                _send(p, out)
            return out
        _send(self.sock, obj)
        return _recv(self.sock)

    def broadcast(self, obj):
        return self.allgather(obj if self.rank == 0 else None)[0]

    def barrier(self):
        self.allgather(0)

    def allgather_bytes(self, data):
        """Byte strings of every rank in rank order (the tiles' way through the host, --allow-host-gather)."""
        return [bytes.fromhex(h) for h in self.allgather(bytes(data).hex())]

    def close(self):
        for p in self.peers:
            p.close()
        if self.sock:
            self.sock.close()
        if self.path:
            try:
                os.remove(self.path)
            except OSError:
                pass
