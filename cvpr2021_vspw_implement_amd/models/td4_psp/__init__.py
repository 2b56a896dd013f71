"""Import surface of the reference drivers (train_clip2.py:18-19, test_clip2.py:15): TDNet (`--method td4_psp`) is
outside the MI355X hot-path scope; its names import and raise at construction."""
