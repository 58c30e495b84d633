"""Optimizing input init points to object surface with the Occupancy Network - MI355X build of ONet/opt_defense.py.

Same flags, defaults and .npz in/out as the reference CLI (ONet/opt_defense.py:21-55,317-387; default config
configs/onet_mn40.yaml, output <dir>/ONet-Opt/onet_opt-<name>):

    python -m ifdefense_amd.onet_opt_defense --data_root=path/to/adv_data.npz --iterations=500
"""
import sys

from .opt_defense import main

if __name__ == '__main__':
    sys.exit(main(model="onet"))
