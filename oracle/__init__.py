"""CPU oracle for the arrow::compute hot path — TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  Nothing under arrow_amd/ does.
"""
