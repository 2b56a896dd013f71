"""`lib.nn` / `lib.utils` of the reference (test_clip2.py:17-18): the three helpers its test driver imports.
(This directory also holds the built libvspw_hip.so.)"""
