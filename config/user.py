"""Per-user settings.  The reference points `bucket` at a GCS URL (/root/reference/config/user.py:1); there is no
network here, so logs go to a local directory unless DDPO_LOGBASE overrides it."""
import os

bucket = os.environ.get("DDPO_LOGBASE", "logs_local")
