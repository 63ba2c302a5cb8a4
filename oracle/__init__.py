"""ORACLE -- test infrastructure only.  See afb_urr_ref.py."""
