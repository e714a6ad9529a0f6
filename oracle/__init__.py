"""oracle/ -- CPU checker for the region-feature path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
package.  The product package (gpt4roi_amd/) never imports it.
"""
