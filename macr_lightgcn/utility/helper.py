"""Small host helpers with the reference's names (macr_lightgcn/utility/helper.py)."""
import os


def ensureDir(dir_path):
    """Create the PARENT directory of dir_path (reference semantics, helper.py:14-17)."""
    d = os.path.dirname(dir_path)
    if d and not os.path.exists(d):
        os.makedirs(d)


def early_stopping(log_value, best_value, stopping_step, expected_order='acc', flag_step=100):
    """helper.py:35-50: `>=` keeps the newer value on ties; returns (best, stopping_step, should_stop)."""
    assert expected_order in ['acc', 'dec']
    better = log_value >= best_value if expected_order == 'acc' else log_value <= best_value
    if better:
        stopping_step, best_value = 0, log_value
    else:
        stopping_step += 1
    should_stop = stopping_step >= flag_step
    if should_stop:
        print("Early stopping is trigger at step: {} log:{}".format(flag_step, log_value))
    return best_value, stopping_step, should_stop
