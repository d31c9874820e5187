"""Pure-Python big-int oracle (spec level).  Test infrastructure only."""
