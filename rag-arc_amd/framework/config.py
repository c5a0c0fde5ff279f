"""Tagged pydantic configs (reference: framework/config.py:11-88).

Every DIRECT subclass must declare, in its own body, ``type: Literal["TAG"] = "TAG"``: the tag is
what discriminated unions dispatch on when a JSON file is turned into a module graph.  Violations
raise TypeError when the class is created; a JSON document carrying another tag fails validation.
"""
from typing import Literal, get_args, get_origin

from pydantic import BaseModel, field_validator


class AbstractConfig(BaseModel):
    def build(self):
        """Return the AbstractModule this config describes."""
        raise NotImplementedError("Subclasses must implement build() method")

    def __init_subclass__(cls, **kwargs):
        super().__init_subclass__(**kwargs)
        own = cls.__dict__.get("__annotations__", {})
        name = cls.__name__
        if "type" not in own:
            raise TypeError(f"{name} must declare `type: Literal['TAG'] = 'TAG'`")
        ann, default = own["type"], cls.__dict__.get("type")
        if isinstance(ann, str):  # postponed annotations: only the spelling can be checked
            if not ann.startswith("Literal["):
                raise TypeError(f"{name}.type must be annotated as Literal['TAG']")
            if default is None:
                raise TypeError(f"{name}.type must have a default value")
            return
        if get_origin(ann) is not Literal:
            raise TypeError(f"{name}.type must be annotated as Literal['TAG']")
        tags = get_args(ann)
        if len(tags) != 1 or not isinstance(tags[0], str):
            raise TypeError(f"{name}.type must be Literal['<single string>']")
        if default != tags[0]:
            raise TypeError(f"{name}.type default must equal {tags[0]!r}")

    @field_validator("type", check_fields=False)
    @classmethod
    def _tag_matches(cls, value):
        expected = cls.__dict__.get("type")
        if "type" in cls.__annotations__ and expected is not None and value != expected:
            raise ValueError(f"type must be {expected!r}")
        return value
