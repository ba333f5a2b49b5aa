"""
Rendezvous of the ranks of one node without any framework: a star of TCP connections on 127.0.0.1 (rank 0 listens,
every other rank holds one socket to it), standard library + numpy only.

It carries what the one-process-per-GPU drivers need *around* the data path -- the 128-byte RCCL unique id, barriers,
the max over the ranks of a timing, a yes/no vote -- and, for rehearsing the sharded factorisation with several ranks on
ONE GPU, the host-staged collectives of ``lgp_comm_init_host`` (broadcast / sum / max of a numpy array).  The data
path of a real multi-GPU run never comes through here: panels travel by RCCL over xGMI (``csrc/dist.hip``).

Every collective is entered by all ranks in the same order (as with any communicator); rank 0 gathers, combines and
answers.  No reference counterpart (the reference is single-process; SURVEY.md section 2, negative inventory).

How the ranks find rank 0's port:
* ``LGP_RDZV_PORT`` (an explicit port: tests), else
* a file ``<tmp>/lgp_rdzv_<uid>/<parent pid>_<MASTER_PORT>`` (directory 0700, file 0600, owner checked) that rank 0 writes after
  binding an ephemeral port -- no port is chosen before it is bound: ``bench.py``'s own launcher and ``python -m
  torch.distributed.run`` both work this way (MASTER_PORT only names the job; under torchrun it belongs to the launcher's store).
The handshake carries a token shared by the ranks of one job (``LGP_RDZV_TOKEN`` from the launcher).
"""

import json
import os
import socket
import struct
import tempfile
import time

import numpy

_MAGIC = b'LGPRDZV1'
OP_BCAST, OP_SUM, OP_MAX, OP_MIN, OP_GATHER, OP_SENDRECV = 0, 1, 2, 3, 4, 5


def free_port(addr='127.0.0.1'):
    """ a port that was free a moment ago (the launcher hands it to the ranks it starts) """
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind((addr, 0))
        return s.getsockname()[1]


def _sendall(sock, payload):
    sock.sendall(struct.pack('<Q', len(payload)))
    if len(payload):
        sock.sendall(payload)


def _recvexact(sock, n):
    buf = bytearray(n)
    view = memoryview(buf)
    got = 0
    while got < n:
        k = sock.recv_into(view[got:], n - got)
        if k == 0:
            raise ConnectionError('rendezvous peer closed the connection')
        got += k
    return buf


def _recvmsg(sock):
    n, = struct.unpack('<Q', _recvexact(sock, 8))
    return _recvexact(sock, n) if n else bytearray()


def _discovery_path():
    """ the file through which rank 0 tells the ranks of its job where it listens: inside a directory that only this user can enter
    (created 0700; refused if it belongs to somebody else or is open to others), so that no other local user can plant or replace it """
    job = os.environ.get('MASTER_PORT', '0')
    uid = os.getuid() if hasattr(os, 'getuid') else 0
    d = os.path.join(tempfile.gettempdir(), f'lgp_rdzv_{uid}')
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    import stat
    if not stat.S_ISDIR(st.st_mode) or (hasattr(os, 'getuid') and st.st_uid != uid) or (st.st_mode & 0o077):
        raise PermissionError(f'rendezvous directory {d} is not a private directory of this user')
    return os.path.join(d, f'{os.getppid()}_{job}')


def _token():
    """ 16 bytes that the ranks of one job share (LGP_RDZV_TOKEN from the launcher; under torchrun its run id) and a stranger does not """
    import hashlib
    secret = os.environ.get('LGP_RDZV_TOKEN') or os.environ.get('TORCHELASTIC_RUN_ID') or ''
    return hashlib.sha256(('lgp-rdzv:' + secret).encode()).digest()[:16]


class Rendezvous:
    """ ``Rendezvous(rank, world)``: connect (blocking, up to ``timeout`` seconds); then ``barrier()``, ``allreduce(x, op)``,
    ``broadcast_bytes(b, src)``, ``collective(op, array, root)`` (in place, the signature ``Context.comm_init_host`` wants). """

    def __init__(self, rank, world, addr=None, port=None, timeout=600.):
        self.rank, self.world = int(rank), int(world)
        if not 0 <= self.rank < self.world:
            raise ValueError(f'rank {rank} outside world of {world}')
        self.addr = addr or os.environ.get('LGP_RDZV_ADDR') or '127.0.0.1'
        self.timeout = float(timeout)
        self._peers = {}          # rank 0: rank -> socket
        self._sock = None         # other ranks: the socket to rank 0
        self._listener = None
        self._path = None
        if self.world == 1:
            return
        if port is None and os.environ.get('LGP_RDZV_PORT'):
            port = int(os.environ['LGP_RDZV_PORT'])
        if self.rank == 0:
            self._serve(port)
        else:
            self._connect(port)

    # ---- set-up
    def _serve(self, port):
        lst = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
        lst.setsockopt(socket.SOL_SOCKET, socket.SO_REUSEADDR, 1)
        lst.bind((self.addr, port or 0))
        lst.listen(self.world)
        self._listener = lst
        if not port:                       # publish the ephemeral port for the ranks that share our parent process
            self._path = _discovery_path()
            tmp = f'{self._path}.{os.getpid()}'
            with open(os.open(tmp, os.O_WRONLY | os.O_CREAT | os.O_EXCL, 0o600), 'w') as f:
                json.dump({'port': lst.getsockname()[1], 'world': self.world, 'pid': os.getpid()}, f)
            os.replace(tmp, self._path)
        deadline = time.monotonic() + self.timeout
        while len(self._peers) < self.world - 1:
            lst.settimeout(max(0.1, deadline - time.monotonic()))
            try:
                conn, _ = lst.accept()
            except socket.timeout:
                missing = sorted(set(range(1, self.world)) - set(self._peers))
                raise TimeoutError(f'rendezvous: ranks {missing} did not connect within {self.timeout:.0f} s') from None
            try:
                conn.settimeout(10.)
                hello = _recvexact(conn, len(_MAGIC) + 8 + 16)
                r, w = struct.unpack('<ii', hello[len(_MAGIC):len(_MAGIC) + 8])
                if (bytes(hello[:len(_MAGIC)]) != _MAGIC or bytes(hello[len(_MAGIC) + 8:]) != _token() or w != self.world or not 0 < r < self.world
                        or r in self._peers):
                    conn.close()
                    continue
                conn.sendall(_MAGIC + _token())
            except (OSError, ConnectionError, struct.error):
                conn.close()
                continue
            conn.settimeout(None)
            conn.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
            self._peers[r] = conn

    def _connect(self, port):
        deadline = time.monotonic() + self.timeout
        path = None if port else _discovery_path()
        last = 'no attempt'
        while time.monotonic() < deadline:
            p = port
            if p is None:
                try:
                    with open(path) as f:
                        if hasattr(os, 'getuid') and os.fstat(f.fileno()).st_uid != os.getuid():
                            raise OSError('the discovery file belongs to another user')
                        d = json.load(f)
                    p = int(d['port']) if int(d.get('world', -1)) == self.world else None
                except (OSError, ValueError, KeyError):
                    p = None
            if p:
                s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
                try:
                    s.settimeout(10.)
                    s.connect((self.addr, p))
                    s.sendall(_MAGIC + struct.pack('<ii', self.rank, self.world) + _token())
                    if bytes(_recvexact(s, len(_MAGIC) + 16)) != _MAGIC + _token():
                        raise ConnectionError('not a rendezvous of this job')
                    s.settimeout(None)
                    s.setsockopt(socket.IPPROTO_TCP, socket.TCP_NODELAY, 1)
                    self._sock = s
                    return
                except (OSError, ConnectionError) as exc:
                    last = repr(exc)
                    s.close()
            time.sleep(0.05)
        raise TimeoutError(f'rendezvous: rank {self.rank} could not reach rank 0 within {self.timeout:.0f} s ({last})')

    # ---- the one primitive: everybody contributes bytes, rank 0 combines, everybody receives bytes
    def _exchange(self, payload, combine):
        if self.world == 1:
            return combine([payload])
        if self.rank == 0:
            parts = [payload] + [_recvmsg(self._peers[r]) for r in range(1, self.world)]
            out = combine(parts)
            if isinstance(out, list):            # one answer per rank
                for r in range(1, self.world):
                    _sendall(self._peers[r], out[r])
                return out[0]
            for r in range(1, self.world):
                _sendall(self._peers[r], out)
            return out
        _sendall(self._sock, payload)
        return _recvmsg(self._sock)

    def barrier(self):
        self._exchange(b'', lambda parts: b'')

    def allgather_bytes(self, b):
        """ list of every rank's bytes, in rank order """
        def combine(parts):
            return b''.join(struct.pack('<Q', len(p)) + bytes(p) for p in parts)
        out = bytes(self._exchange(bytes(b), combine))
        res, off = [], 0
        while off < len(out):
            n, = struct.unpack_from('<Q', out, off)
            res.append(out[off + 8:off + 8 + n])
            off += 8 + n
        return res

    def broadcast_bytes(self, b, src=0):
        return self.allgather_bytes(b if self.rank == src else b'')[src]

    def allreduce(self, x, op='max'):
        """ float or array; op in 'sum', 'max', 'min' """
        scalar = numpy.ndim(x) == 0
        a = numpy.array(x, dtype=numpy.float64, ndmin=1, order='C')        # a copy: the caller's value is left alone
        self.collective({'sum': OP_SUM, 'max': OP_MAX, 'min': OP_MIN}[op], a, 0)
        return float(a[0]) if scalar else a

    def collective(self, op, arr, root):
        """ in place on the contiguous numpy array ``arr``: op 0 broadcast from ``root``, 1 sum, 2 max, 3 min.  The reductions are
        done in rank order on rank 0, so every rank receives the same bits.  op 5: the bytes of rank ``root & 0xffff`` to rank
        ``root >> 16`` (every rank calls; the others' arrays are left alone). """
        if not arr.flags.c_contiguous or not arr.flags.writeable:
            raise ValueError('collective needs a writeable contiguous array (it is modified in place)')
        flat = arr.reshape(-1)           # a view
        dt = flat.dtype
        if op == OP_SENDRECV:
            src, dst = root & 0xffff, root >> 16
            if not (0 <= src < self.world and 0 <= dst < self.world) or src == dst:
                raise ValueError(f'send/receive between ranks {src} and {dst} of {self.world}')
            out = self._exchange(flat.tobytes() if self.rank == src else b'',
                                 lambda parts: [parts[src] if r == dst else b'' for r in range(self.world)])
            if self.rank == dst:
                got = numpy.frombuffer(out, dtype=dt)
                if got.size != flat.size:
                    raise ValueError(f'send/receive: {got.size} elements arrived, {flat.size} expected')
                flat[...] = got
            return

        def combine(parts):
            if op == OP_BCAST:
                return parts[root]
            acc = numpy.frombuffer(parts[0], dtype=dt).copy()
            for p in parts[1:]:
                v = numpy.frombuffer(p, dtype=dt)
                if op == OP_SUM:
                    acc += v
                elif op == OP_MAX:
                    numpy.maximum(acc, v, out=acc)
                else:
                    numpy.minimum(acc, v, out=acc)
            return acc.tobytes()
        mine = flat.tobytes() if (op != OP_BCAST or self.rank == root) else b''
        out = self._exchange(mine, combine)
        if not (op == OP_BCAST and self.rank == root):
            flat[...] = numpy.frombuffer(out, dtype=dt)

    def close(self):
        for s in list(self._peers.values()) + [self._sock, self._listener]:
            if s is not None:
                try:
                    s.close()
                except OSError:
                    pass
        self._peers, self._sock, self._listener = {}, None, None
        if self._path:
            try:
                os.unlink(self._path)
            except OSError:
                pass
            self._path = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
