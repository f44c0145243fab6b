"""Device plumbing: PyTorch-ROCm is used ONLY for device memory, streams and torch.distributed.  All arithmetic on
event data happens in libevk.so."""
import ctypes

import numpy as np
import torch

from . import _lib


def require_gpu():
    if not torch.cuda.is_available():
        raise _lib.EvkError("event_utils_amd needs an AMD GPU (MI355X / gfx950): torch.cuda.is_available() is False "
                            "and there is no CPU fallback")
    _lib.lib()
    return torch.device("cuda", torch.cuda.current_device())


try:                                   # fast path to the current stream handle (what torch.compile / triton use)
    _raw_stream = torch._C._cuda_getCurrentRawStream
except AttributeError:                 # pragma: no cover
    _raw_stream = None


def stream_id(device=None):
    """hipStream_t of torch's current stream on `device` (int)."""
    if _raw_stream is not None:
        idx = torch.cuda.current_device() if device is None or device.index is None else device.index
        return _raw_stream(idx)
    return torch.cuda.current_stream(device).cuda_stream


def stream():
    return ctypes.c_void_p(stream_id())


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    if t is None:
        return None
    return ctypes.c_void_p(t.data_ptr())


def host_ptr(a):
    return ctypes.c_void_p(a.ctypes.data)


_NP2T = {np.dtype(np.float32): torch.float32, np.dtype(np.float64): torch.float64,
         np.dtype(np.int32): torch.int32, np.dtype(np.int64): torch.int64}


_RAW_UPLOAD = (np.float64, np.float32, np.int64, np.int32, np.int16, np.int8, np.uint8, np.bool_)


def to_device(a, dtype, device=None):
    """numpy array / torch tensor (any device) -> contiguous 1-D+ torch tensor of `dtype` on the GPU.
    No copy when `a` already is such a tensor."""
    if isinstance(a, torch.Tensor):
        if a.is_cuda and a.dtype == dtype and a.is_contiguous() and (device is None or a.device == device):
            return a                    # (the common case of a public call on resident columns: no dispatcher round trip)
        device = device or require_gpu()
        return a.to(device=device, dtype=dtype, non_blocking=True).contiguous()
    device = device or require_gpu()
    a = np.asarray(a)
    want = {torch.float32: np.float32, torch.float64: np.float64, torch.int32: np.int32, torch.int64: np.int64}[dtype]
    if a.dtype != want:
        # (round 6) a column of another width goes up AS IT IS and is converted by a device copy: numpy's single-threaded astype
        # of 10 M int64 / float64 values takes 4.6 ms and the upload of its result 1.5 ms more, the raw upload + device
        # conversion 1.5 ms together.  Same values: C conversions between these types round / wrap alike on both sides.
        if a.dtype.type in _RAW_UPLOAD and a.dtype.isnative and a.size and a.flags.writeable:
            return torch.from_numpy(np.ascontiguousarray(a)).to(device, non_blocking=False).to(dtype)
        a = a.astype(want)
    a = np.ascontiguousarray(a)
    return torch.from_numpy(a).to(device, non_blocking=False)


def ends(col):
    """(col[0], col[-1]) of a device column as python floats with ONE device-to-host transfer (each .item() is a
    synchronisation of its own); a host column is read directly."""
    if col.is_cuda:
        a, b = torch.stack((col[0], col[-1])).tolist()
        return float(a), float(b)
    return float(col[0]), float(col[-1])


def zeros(shape, dtype=torch.float32, device=None):
    return torch.zeros(shape, dtype=dtype, device=device or require_gpu())


def error_mode():
    """'strict' (default since round 5: the reference's behaviour, image.py:96-99): a call that dropped out-of-range
    events raises before it returns.  On the one-pass paths that costs no stream synchronisation: the call waits for its
    partition kernel's own report in a pinned slot while the tile kernel runs on (0.0728 against 0.0721 ms per 10 M-event
    call); the direct kernels synchronise the stream.  'deferred' (EVK_ERRORS=deferred, opt-in): calls whose inputs AND
    outputs stay on the device only enqueue; the exception surfaces at the next event_utils_amd call on that stream or
    at check_errors() -- the way the reference's own CUDA path reports an out-of-range index_put_ (an asynchronous
    device-side assert).  Calls that hand their result to the host are always strict (they synchronise anyway)."""
    return _lib.getenv("EVK_ERRORS", "strict")


class _ErrorState:
    """Per (device, stream): ONE cumulative device counter of dropped events (never reset, so no memset per call).
    Deferred reports reach the host without a copy or an event on the stream: the kernels of a call write
    {call sequence number, counter} into a pinned, device-visible slot when their last workgroup finishes
    (evk_voxel2_f32, `host_report`); calls whose kernels cannot do that copy the counter into a ring of pinned slots
    asynchronously and record an event."""
    RING = 64

    def __init__(self, device):
        from collections import deque
        self.counter = torch.zeros(1, dtype=torch.int32, device=device)
        self.host = torch.zeros(self.RING, dtype=torch.int32).pin_memory()
        self.report = torch.zeros(2, dtype=torch.int32).pin_memory()      # [sequence number, counter], written by the GPU
        self.report_np = self.report.numpy()
        self.seq = 0
        self.seen = 0
        self.slot = 0
        self.pending = deque()

    def _advance(self, value):
        """Events dropped since the last report, given a snapshot `value` of the cumulative counter.  A snapshot OLDER than
        what was already reported (a deferred entry polled after a later strict call) counts as 0: `seen` never moves
        backwards, so nothing is reported twice and no wrapped difference masquerades as ~4e9 events."""
        n = (value - self.seen) & 0xFFFFFFFF
        if n >= 0x80000000:
            return 0
        self.seen = value & 0xFFFFFFFF
        return n

    def _raise(self, value, exc_type, msg):
        n = self._advance(value)
        if n:
            raise exc_type("%s (%d offending events)" % (msg, n))

    def drain(self):
        """Wait for every pending deferred report and fold it into `seen` WITHOUT raising -> events they dropped.  For callers
        that must not raise on one rank alone (distributed.py: the count is summed over the ranks and all raise)."""
        if not self.pending:
            return 0
        torch.cuda.synchronize()
        n = 0
        while self.pending:
            ev, k, _, _ = self.pending.popleft()
            n += self._advance((int(self.report_np[1]) if ev is None else int(self.host[k])) & 0xFFFFFFFF)
        return n

    def poll(self, wait=False):
        while self.pending:
            ev, k, exc_type, msg = self.pending[0]
            if ev is None:                       # reported by the kernels: k = the call's sequence number
                if wait:
                    torch.cuda.synchronize()
                done, value = int(self.report_np[0]), int(self.report_np[1])
                if ((done - k) & 0xFFFFFFFF) >= 0x80000000:      # that call has not finished yet
                    return
                self.pending.popleft()
                self._raise(value & 0xFFFFFFFF, exc_type, msg)
                continue
            if wait:
                ev.synchronize()
            elif not ev.query():
                return
            self.pending.popleft()
            self._raise(int(self.host[k]), exc_type, msg)

    def strict(self, exc_type, msg):
        self.poll(wait=True)
        self._raise(int(self.counter.item()), exc_type, msg)      # synchronises; the reference is synchronous too

    def strict_report(self, seq, exc_type, msg):
        """'strict' for a call whose kernels report {sequence number, counter} to the pinned slot themselves (the one-pass
        paths: written when the partition kernel's last workgroup finishes; the tile kernel behind it drops nothing): wait for
        THAT report instead of synchronising the stream and reading the counter back -- the exception is as synchronous as
        the reference's, the tile kernel keeps running and the next call's launches are not held up (10 M events: 0.106 ->
        ~0.08 ms per call).  A stream that went idle without reporting (a failed launch) falls back to the read-back."""
        if self.pending:
            self.poll(wait=True)
        spins = 0
        while ((int(self.report_np[0]) - seq) & 0xFFFFFFFF) >= 0x80000000:
            spins += 1
            if spins % 2048 == 0 and torch.cuda.current_stream(self.counter.device).query():
                if ((int(self.report_np[0]) - seq) & 0xFFFFFFFF) >= 0x80000000:
                    return self.strict(exc_type, msg)
        self._raise(int(self.report_np[1]) & 0xFFFFFFFF, exc_type, msg)

    def next_seq(self):
        self.seq = (self.seq + 1) & 0x7FFFFFFF
        return self.seq

    def defer(self, exc_type, msg, seq=None):
        if len(self.pending) >= self.RING - 1:
            self.poll(wait=True)
        if seq is not None:
            self.pending.append((None, seq, exc_type, msg))
            return
        k, self.slot = self.slot, (self.slot + 1) % self.RING
        self.host[k:k + 1].copy_(self.counter, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        self.pending.append((ev, k, exc_type, msg))


_errors = {}


def _error_state(device):
    key = (device.index, stream_id(device))
    st = _errors.get(key)
    if st is None:
        st = _errors[key] = _ErrorState(device)
    return st


def check_errors():
    """Synchronise and raise the exception of any earlier deferred call that dropped out-of-range events."""
    torch.cuda.synchronize()
    for st in list(_errors.values()):
        st.poll(wait=True)


def _report_at_exit():
    """A deferred report that nobody collected (no later event_utils_amd call, no check_errors()) must not vanish with the
    process: say so on stderr.  (An exception cannot propagate out of an atexit hook.)"""
    import sys
    try:
        if not any(st.pending for st in _errors.values()):    # nothing deferred: do not touch (or initialise) the device
            return
        check_errors()
    except (IndexError, ValueError) as e:
        sys.stderr.write("event_utils_amd: an earlier call dropped out-of-range events and the error was never collected "
                         "(EVK_ERRORS=deferred; call check_errors() or set EVK_ERRORS=strict): %s\n" % (e,))
    except Exception:   # noqa: BLE001  (interpreter shutdown: the device may already be gone)
        pass


import atexit  # noqa: E402
atexit.register(_report_at_exit)


class OobCounter:
    """Device counter of events the reference would have rejected with an exception (see error_mode)."""

    def __init__(self, device=None, poll=True):
        self.state = _error_state(device or require_gpu())
        if poll:                               # surface what earlier deferred calls on this stream left behind
            self.state.poll()                  # (poll=False: sharded callers, which must raise on every rank together)
        self.seq = None                        # set by a call whose kernels report {seq, counter} to the host themselves

    def report_args(self):
        """(pinned slot pointer, sequence number) for kernels that report the counter to the host themselves."""
        self.seq = self.state.next_seq()
        return ctypes.c_void_p(self.state.report.data_ptr()), self.seq

    @property
    def ptr(self):
        return ptr(self.state.counter)

    def raise_if_set(self, exc_type, msg, deferrable=False):
        if deferrable and error_mode() == "deferred":
            self.state.defer(exc_type, msg, self.seq)
        elif deferrable and self.seq is not None:
            self.state.strict_report(self.seq, exc_type, msg)     # (results stay on the device: only the REPORT is waited for)
        else:
            self.state.strict(exc_type, msg)


_scratch = {}


def reduce_scratch(device):
    key = (device.index, stream_id(device), "reduce")
    if key not in _scratch:
        nbytes = int(_lib.lib().evk_reduce_scratch_bytes())
        _scratch[key] = (torch.empty(nbytes // 8, dtype=torch.float64, device=device), nbytes)
    return _scratch[key]


def out4(device, n=4):
    """Persistent n-double device result slot (objective evaluations return a few scalars)."""
    key = (device.index, stream_id(device), "out%d" % n)
    if key not in _scratch:
        _scratch[key] = torch.empty(n, dtype=torch.float64, device=device)
    return _scratch[key]
