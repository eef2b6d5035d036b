import sys, os
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from ptina_amd import scenes, common
from ptina_amd.common import ctx
from helpers import setup_engine
eng = setup_engine(scenes.get_scene('s978'), 512, 512, mode='fast')
c = ctx()
c.set_option('batch', 32)
eng.render(1); c.call('mpt_synchronize')
c.set_option('count', 1); c.call('mpt_reset_counters')
eng.render(32)
cnt = c.counters()
print('leaf steps', cnt['pl_trips'], 'followed by a leaf', cnt['pl_local'], cnt['pl_local'] / max(cnt['pl_trips'], 1), 'kernel', c.get_option('last_kernel'))
common.reset_all()
