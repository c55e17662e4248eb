"""Stand-in for `ray` (optimizer.py:16, utils/task_pool.py import it at module level).  TEST INFRASTRUCTURE ONLY.
Only the single-process optimizer (optimizer.py:286-397) is on the path; it never touches Ray, so every attribute raises."""


def __getattr__(name):
    raise NotImplementedError('ray.%s: the Ray actor plumbing is out of scope (DESIGN.md section 8)' % name)
