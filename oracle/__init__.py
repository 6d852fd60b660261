"""ORACLE: CPU restatements of the reference's ENF decoder (test infrastructure only).

PARITY UNPINNED -- see enf_ref_np.py.  Nothing under enf-pde_amd/ may import this package.
"""
