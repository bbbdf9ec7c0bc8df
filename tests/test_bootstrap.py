"""xmhw_amd/bootstrap.py: which interface rank 0 listens on (a host name that /etc/hosts maps to 127.0.1.1 --
the Debian / Ubuntu default for a machine's own name -- must not bind the loopback, or the ranks of other nodes, which
resolve the routable address, are refused), and the payload exchange itself on the loopback."""
import socket
import threading

from xmhw_amd import bootstrap


def test_bind_address(monkeypatch):
    assert bootstrap._bind_address("127.0.0.1") == "127.0.0.1"          # a literal address of this host: exactly that interface
    #  a literal that this host does not own -- a NAT / floating address, a service VIP, an address seen through
    # another network namespace -- must not be bound (rank 0 would die with "Cannot assign requested address" and the
    # other ranks would spin until the timeout): every interface.  The same for an IPv6 literal on the IPv4 socket.
    assert bootstrap._bind_address("203.0.113.7") == "0.0.0.0"
    assert bootstrap._bind_address("::1") == "0.0.0.0"
    assert bootstrap._bind_address("2001:db8::1") == "0.0.0.0"
    assert bootstrap._bind_address("localhost") == "127.0.0.1"
    monkeypatch.setattr(socket, "gethostbyname", lambda name: "127.0.1.1")
    assert bootstrap._bind_address("node0") == "0.0.0.0"                # own name -> loopback alias: every interface
    monkeypatch.setattr(socket, "gethostbyname", lambda name: "203.0.113.7")
    assert bootstrap._bind_address("elsewhere") == "0.0.0.0"            # resolves, but not an address of this host

    def boom(name):
        raise OSError("no such host")
    monkeypatch.setattr(socket, "gethostbyname", boom)
    assert bootstrap._bind_address("unknown-host") == "0.0.0.0"


def test_share_bytes_with_a_rendezvous_address_this_host_does_not_own():
    """rank 0 is told a literal it cannot bind (203.0.113.7, TEST-NET-3); it listens on every interface and a rank that
    reaches it another way (here the loopback) is served"""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    got = {}

    def rank0():
        got[0] = bootstrap.share_bytes(0, 2, lambda: b"id", addr="203.0.113.7", port=port, timeout=20.0)

    def rank1():
        got[1] = bootstrap.share_bytes(1, 2, lambda: b"", addr="127.0.0.1", port=port, timeout=20.0)
    ts = [threading.Thread(target=f) for f in (rank0, rank1)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(30)
    assert got == {0: b"id", 1: b"id"}


def test_share_bytes_two_ranks_on_the_loopback():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    got = {}

    def rank(r):
        got[r] = bootstrap.share_bytes(r, 2, lambda: b"unique-id-bytes", addr="127.0.0.1", port=port, timeout=20.0)
    ts = [threading.Thread(target=rank, args=(r,)) for r in (0, 1)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(30)
    assert got == {0: b"unique-id-bytes", 1: b"unique-id-bytes"}
