"""Stand-in for the absent `statsmodels` (TEST INFRASTRUCTURE, like refshim/numba): the reference's
Visualization/utils.py imports statsmodels.sandbox.distributions.extras at module level but only
generate_normal_four_moments (not part of the pinned posterior core) calls into it."""
