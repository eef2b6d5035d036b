'''
run every call of a module on one daemon thread (reference tools/mtworker.py:22-89): the Blender
add-on drives ptina.worker through such a proxy because the device runtime must be used from a
single thread -- libmiptina.so has the same contract (include/miptina.h: a context is not
re-entrant).  Exceptions in the worker are printed and the call returns None, as in the reference.
'''

import queue
import threading
import traceback


class DaemonWorker:
    def __init__(self):
        self.q = queue.Queue(maxsize=4)
        self.t = threading.Thread(target=self._main, daemon=True)
        self.t.start()

    def _main(self):
        while True:
            func, box, done = self.q.get()
            try:
                box.append(func())
            except Exception:
                traceback.print_exc()
                box.append(None)
            finally:
                done.set()

    def launch(self, func):
        box, done = [], threading.Event()
        self.q.put((func, box, done))
        return box, done

    def wait_done(self, handle):
        box, done = handle
        done.wait()
        return box[0]


class DaemonModule:
    '''proxy: attribute access returns wrappers that run the module's function on the worker'''

    def __init__(self, getmodule):
        self._worker = DaemonWorker()
        self._getmodule = getmodule
        self._module = None

    def __getattr__(self, name):
        def call(*args, **kwargs):
            def job():
                if self._module is None:
                    self._module = self._getmodule()
                return getattr(self._module, name)(*args, **kwargs)
            return self._worker.wait_done(self._worker.launch(job))
        call.__name__ = name
        return call


class OnDemandProxy:
    '''build the wrapped object at first use (reference mtworker.py:75-89)'''

    def __init__(self, getobject):
        self._getobject = getobject
        self._object = None

    def __getattr__(self, name):
        if self._object is None:
            self._object = self._getobject()
        return getattr(self._object, name)
