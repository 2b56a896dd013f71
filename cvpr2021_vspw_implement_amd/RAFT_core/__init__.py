"""Import-surface mirror of the reference's top-level RAFT_core package (models/netwarp.py:9-10 imports
`RAFT_core.raft.RAFT` and `RAFT_core.utils.utils.InputPadder`)."""
