'''
`import ptina...` -> ptina_amd.  Lets a PTina driver script (e.g. the reference's exams/benchmark.py)
run against the MI355X implementation with its import lines untouched: every `ptina.X` module is the
module `ptina_amd.X`.  Nothing of the reference is in here.
'''

import importlib
import importlib.abc
import importlib.util
import sys

import ptina_amd


class _AliasFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    prefix = __name__ + '.'

    def find_spec(self, fullname, path=None, target=None):
        if not fullname.startswith(self.prefix):
            return None
        real = 'ptina_amd.' + fullname[len(self.prefix):]
        if importlib.util.find_spec(real) is None:
            return None
        return importlib.util.spec_from_loader(fullname, self, is_package=hasattr(importlib.import_module(real), '__path__'))

    def create_module(self, spec):
        return importlib.import_module('ptina_amd.' + spec.name[len(self.prefix):])

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, _AliasFinder())
__path__ = list(ptina_amd.__path__)
