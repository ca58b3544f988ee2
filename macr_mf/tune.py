"""c-sweep tuner -- drop-in for the reference's macr_mf/tune.py (a copy of train.py whose evaluation loops
over np.linspace(--start, --end, --step) values of c, tune.py:545-578).  c only enters the evaluator's
epilogue, so each extra value costs one more fused scoring/top-K pass (milliseconds).

    python ./macr_mf/tune.py --dataset ml_10m/val --batch_size 8192 --start 30 --end 40 --step 11 \
        --train rubibceboth --test rubi --alpha 1e-3 --beta 1e-3 --valid_set valid
"""
from train import main

if __name__ == '__main__':
    main(sweep=True)
