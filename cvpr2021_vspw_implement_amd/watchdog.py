"""Phase log + hang watchdog for the multi-rank entry points (bench.py --gpus N, train_clip2 under torchrun, the
two-rank test workers).

A collective that never completes blocks the calling thread inside C++ with the GIL released, so a Python-level
watchdog thread keeps running: when no `phase()` call has arrived for `limit_s` seconds it dumps every thread's Python
stack (faulthandler) to stderr, calls `on_expire` (bench.py prints its one JSON line with an "error" key there) and
ends the process with exit code 3 - a hung rank becomes a failed run with a stack in the log instead of a silent
timeout of whoever launched it.  No reference counterpart (the reference is one process, nn.DataParallel threads).
"""
import faulthandler
import os
import sys
import threading
import time


class Watchdog:
    def __init__(self, limit_s=120.0, on_expire=None, tag=None, verbose=True):
        self.limit = float(limit_s)
        self.on_expire = on_expire
        self.tag = tag if tag is not None else "rank %s" % os.environ.get("RANK", "0")
        self.verbose = verbose
        self.t_start = time.time()
        self.t_last = self.t_start
        self.name = "start"
        self.limit_now = self.limit
        self._stop = threading.Event()
        self._thread = threading.Thread(target=self._run, name="vspw-watchdog", daemon=True)
        self._thread.start()

    def phase(self, name, limit_s=None):
        """Mark progress: the phase `name` begins now and may take up to limit_s (default: the watchdog's limit)."""
        now = time.time()
        if self.verbose:
            sys.stderr.write("[%s +%7.2fs] %s\n" % (self.tag, now - self.t_start, name))
            sys.stderr.flush()
        self.name = name
        self.limit_now = self.limit if limit_s is None else float(limit_s)
        self.t_last = now

    def stop(self):
        self._stop.set()

    def _run(self):
        while not self._stop.wait(1.0):
            if time.time() - self.t_last > self.limit_now:
                sys.stderr.write("[%s] WATCHDOG: phase %r exceeded %.0f s - dumping stacks and exiting\n"
                                 % (self.tag, self.name, self.limit_now))
                sys.stderr.flush()
                try:
                    faulthandler.dump_traceback(file=sys.stderr, all_threads=True)
                except Exception:  # noqa: BLE001
                    pass
                try:
                    if self.on_expire is not None:
                        self.on_expire(self.name)
                except Exception:  # noqa: BLE001
                    pass
                sys.stderr.flush()
                sys.stdout.flush()
                os._exit(3)


class _Null:
    def phase(self, name, limit_s=None):
        pass

    def stop(self):
        pass


def make(enabled, limit_s=120.0, on_expire=None, tag=None, verbose=True):
    """Watchdog when enabled, a no-op object with the same interface otherwise."""
    if not enabled:
        return _Null()
    limit_s = float(os.environ.get("VSPW_WATCHDOG_S", limit_s))
    return Watchdog(limit_s, on_expire, tag, verbose)
