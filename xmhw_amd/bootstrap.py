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


def _is_local_v4(ip):
    """can a server socket of this host bind that IPv4 address?"""
    probe = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    try:
        probe.bind((ip, 0))
        return True
    except OSError:
        return False
    finally:
        probe.close()


def _bind_address(addr):
    """The interface rank 0 listens on.  The rendezvous address itself when this host owns it -- a literal IPv4
    address, or a HOST NAME that resolves to a non-loopback address of this host (Debian / Ubuntu map a machine's own
    name to 127.0.1.1 in /etc/hosts, and a server bound there would refuse the ranks of the other nodes, which resolve
    the routable address); the loopback for `localhost` / 127.x (a single-node job stays off the network); otherwise
    every interface: a literal that is NOT an address of this host (a NAT or floating address, a service VIP, an
    address seen through another network namespace), an IPv6 literal (the server socket is IPv4), a name that does
    not resolve."""
    import ipaddress
    try:
        ip = ipaddress.ip_address(addr)
        if ip.version == 4 and _is_local_v4(addr):
            return addr                                  # a literal address of this host: exactly that interface
        return "0.0.0.0"
    except ValueError:
        pass
    if addr == "localhost":
        return "127.0.0.1"
    try:
        resolved = socket.gethostbyname(addr)
        if not ipaddress.ip_address(resolved).is_loopback and _is_local_v4(resolved):
            return resolved
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
        try:
            srv.bind((_bind_address(addr), port))
        except OSError:
            srv.bind(("0.0.0.0", port))                  # (the probe and the bind raced, or the port is per-interface)
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
