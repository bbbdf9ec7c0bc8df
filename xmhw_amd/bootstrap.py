"""Out-of-band exchange of a few bytes between the ranks of a sharded run (the 128-byte RCCL id).

RCCL needs every rank to call ``ncclCommInitRank`` with the SAME unique id, created on one rank.
How the id travels is not RCCL's business; here rank 0 listens on a TCP port and hands the payload
to the other ranks -- the same star a process launcher's rendezvous uses, in ~40 lines of sockets,
so that the product path needs neither MPI nor PyTorch.  On one node the address is 127.0.0.1.
"""
import os
import socket
import struct
import time

_MAGIC = b"XMHW"


def default_port():
    if os.environ.get("XMHW_BOOTSTRAP_PORT"):
        return int(os.environ["XMHW_BOOTSTRAP_PORT"])
    # next to the launcher's own rendezvous port (torchrun hosts a store ON MASTER_PORT, so not that one)
    return (int(os.environ.get("MASTER_PORT", "29400")) + 17) % 65536 or 29417


def _recv_exact(conn, n):
    buf = b""
    while len(buf) < n:
        chunk = conn.recv(n - len(buf))
        if not chunk:
            raise ConnectionError("bootstrap peer closed the connection")
        buf += chunk
    return buf


def _bind_address(addr):
    """The interface rank 0 listens on.  The rendezvous address itself when it is a literal IP or plainly the
    loopback (`localhost`, 127.x: a single-node job stays off the network); for a HOST NAME only if it resolves to
    a non-loopback address -- Debian / Ubuntu map a machine's own name to 127.0.1.1 in /etc/hosts, and a server
    bound there would refuse the ranks of the other nodes, which resolve the routable address -- otherwise every
    interface."""
    import ipaddress
    try:
        ipaddress.ip_address(addr)
        return addr                                      # a literal address: exactly that interface
    except ValueError:
        pass
    if addr == "localhost":
        return "127.0.0.1"
    try:
        resolved = socket.gethostbyname(addr)
        if not ipaddress.ip_address(resolved).is_loopback:
            probe = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
            try:
                probe.bind((resolved, 0))                # an address of this host?
                return resolved
            except OSError:
                pass
            finally:
                probe.close()
    except OSError:
        pass
    return "0.0.0.0"


def share_bytes(rank, size, make_payload, addr=None, port=None, timeout=120.0):
    """Rank 0 calls ``make_payload()`` and every rank returns those bytes."""
    if size == 1:
        return make_payload()
    addr = addr or os.environ.get("MASTER_ADDR", "127.0.0.1")
    port = int(port) if port else default_port()
    if rank == 0:
        payload = make_payload()
        srv = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        srv.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        srv.bind((_bind_address(addr), port))
        srv.listen(size)
        srv.settimeout(timeout)
        served = 0
        try:
            while served < size - 1:
                conn, _ = srv.accept()
                with conn:
                    conn.settimeout(5.0)                 # a silent stray connection does not block the group
                    try:
                        if _recv_exact(conn, 4) != _MAGIC:
                            continue
                    except (socket.timeout, OSError):
                        continue
                    conn.sendall(struct.pack("<I", len(payload)) + payload)
                    served += 1
        finally:
            srv.close()
        return payload
    deadline = time.monotonic() + timeout
    while True:
        try:
            with socket.create_connection((addr, port), timeout=5.0) as conn:
                conn.sendall(_MAGIC)
                (n,) = struct.unpack("<I", _recv_exact(conn, 4))
                return _recv_exact(conn, n)
        except (ConnectionRefusedError, ConnectionResetError, socket.timeout, OSError):
            if time.monotonic() > deadline:
                raise
            time.sleep(0.05)
