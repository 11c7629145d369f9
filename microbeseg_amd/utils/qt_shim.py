"""Qt-free stand-ins for the few PyQt5 names the reference workers use (QObject, pyqtSignal, pyqtSlot,
QCoreApplication.processEvents) so that TrainWorker / InferWorker run headless (CLI scripts, tests, GPU box).
If PyQt5 is importable the real classes are used and the GUI (src/microbe_seg_gui.py) can drive the workers
unchanged (signals finished / progress / text_output / text_output_main_gui, reference train.py:117-121)."""
try:  # pragma: no cover - PyQt5 is not part of the build/GPU image
    from PyQt5.QtCore import QObject, pyqtSignal, pyqtSlot, QCoreApplication
    HAVE_QT = True
except Exception:
    HAVE_QT = False

    class _BoundSignal:
        def __init__(self):
            self._slots = []

        def connect(self, fn):
            self._slots.append(fn)

        def disconnect(self, fn=None):
            self._slots = [s for s in self._slots if fn is not None and s is not fn]

        def emit(self, *args):
            for s in list(self._slots):
                s(*args)

    class pyqtSignal:
        """Descriptor creating one bound signal per instance (like Qt does)."""

        def __init__(self, *types):
            self._name = None

        def __set_name__(self, owner, name):
            self._name = "_sig_" + name

        def __get__(self, obj, objtype=None):
            if obj is None:
                return self
            sig = obj.__dict__.get(self._name)
            if sig is None:
                sig = _BoundSignal()
                obj.__dict__[self._name] = sig
            return sig

    def pyqtSlot(*_a, **_k):
        def deco(fn):
            return fn
        return deco

    class QObject:
        def __init__(self, *a, **k):
            super().__init__()

    class QCoreApplication:
        @staticmethod
        def processEvents():
            return None
