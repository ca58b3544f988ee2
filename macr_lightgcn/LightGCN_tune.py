"""c-sweep tuner -- drop-in for the reference's macr_lightgcn/LightGCN_tune.py (LightGCN.py whose evaluation
loops over np.linspace(--start, --end, --step) values of c, LightGCN_tune.py:852-870)."""
from LightGCN import main

if __name__ == '__main__':
    main(sweep=True)
