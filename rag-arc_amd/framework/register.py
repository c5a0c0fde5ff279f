"""Process-wide registry of built applications (reference: framework/register.py:7-26).

register(): a missing file raises FileNotFoundError (open() is outside the guard); unreadable JSON
or a config that fails validation is REPORTED with print() and leaves the registry untouched;
registering an existing name overwrites it.  get_object(): KeyError for unknown names.
"""
import json

from .singleton import singleton


@singleton
class Register:
    def __init__(self):
        self.registrations = {}

    def register(self, config_path, app_name, config_type):
        with open(config_path, "r") as fh:
            try:
                self.registrations[app_name] = config_type(**json.loads(fh.read())).build()
            except Exception as exc:  # noqa: BLE001 - same contract as the reference
                print(f"Error registering {app_name}, the config file is not valid\n {exc}")

    def get_object(self, app_name):
        return self.registrations[app_name]
