import sys, os
sys.path.insert(0, os.getcwd())
sys.argv = ['bench.py', '--mode', 'train', '--steps', '3', '--warmup', '1', '--batch', '8',
            '--train-dtype', os.environ.get('BRCNN_DTYPE', 'f32')]
import bench
bench.main()
