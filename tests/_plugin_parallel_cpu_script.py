"""Child script of test_host_logic.py::test_plugin_cpu_parallel_goes_through_the_reference_fork: a user script with
the reference's YAML defaults (`device: cpu`) and `train_cfgs.parallel = 2`.  The reference re-launches it under
torchrun (utils/distributed.py:83-139, called from algo_wrapper.py:152-157 BEFORE the registry lookup), so with the
plugin installed both workers must still build the reference's own class."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'oracle'))
import ref_harness  # noqa: E402

omnisafe = ref_harness.import_reference()
ref_harness.import_simple_env()
if os.environ.get('WITH_PLUGIN') == '1':
    import omnisafe_amd

    omnisafe_amd.install()
cfg = {'train_cfgs': {'total_steps': 800, 'vector_env_nums': 1, 'torch_threads': 1, 'parallel': 2},
       'algo_cfgs': {'steps_per_epoch': 400, 'update_iters': 2},
       'logger_cfgs': {'use_wandb': False, 'use_tensorboard': False, 'log_dir': sys.argv[1]}}
agent = omnisafe.Agent('PPOLag', 'Test-v0', custom_cfgs=cfg)
print('CLASS', type(agent.agent).__module__, 'RANK', os.environ.get('RANK'), flush=True)
agent.learn()
