"""KBHit -- non-blocking keyboard poll for the p/c/q controls of Learner._learn (reference: util/kbhit.py).
Headless-safe: without a TTY (batch jobs, the GPU box) it reports no key presses instead of raising."""
import os
import sys

try:
    import termios
    from select import select
except ImportError:  # pragma: no cover - non-posix
    termios = None


class KBHit:
    def __init__(self):
        self.enabled = False
        if termios is None or not sys.stdin.isatty():
            return
        try:
            self.fd = sys.stdin.fileno()
            self.old_term = termios.tcgetattr(self.fd)
            new_term = termios.tcgetattr(self.fd)
            new_term[3] = new_term[3] & ~termios.ICANON & ~termios.ECHO
            termios.tcsetattr(self.fd, termios.TCSAFLUSH, new_term)
            import atexit
            atexit.register(self.set_normal_term)
            self.enabled = True
        except Exception:
            self.enabled = False

    def set_normal_term(self):
        if self.enabled:
            termios.tcsetattr(self.fd, termios.TCSAFLUSH, self.old_term)

    def getch(self):
        return sys.stdin.read(1) if self.enabled else ""

    def kbhit(self):
        if not self.enabled:
            return False
        dr, _, _ = select([sys.stdin], [], [], 0)
        return dr != []
