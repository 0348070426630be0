"""CPU parity oracle (TEST INFRASTRUCTURE).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product (marl_dmfb_amd/) never does: it fails loudly when its HIP
extension is missing instead of falling back to this code.
"""
