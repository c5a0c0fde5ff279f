"""The reference's 21 framework tests (framework/{config,module,register}_test.py), restated against
this package's registry: same behaviours, own fixtures."""
import json
from dataclasses import dataclass
from typing import Annotated, List, Literal, Union
from unittest.mock import patch

import pytest
from pydantic import Field, ValidationError

from rag_arc_amd.framework import AbstractConfig, AbstractModule, Register
from rag_arc_amd.framework.singleton import singleton


# ---------------------------------------------------------------- fixtures: a small config family
class LeafAConfig(AbstractConfig):
    type: Literal["A"] = "A"
    value: int = 1

    def build(self):
        return LeafA(config=self)


class LeafCConfig(AbstractConfig):
    type: Literal["C"] = "C"
    name: str = "c"

    def build(self):
        return LeafC(config=self)


@dataclass
class LeafA(AbstractModule):
    config: LeafAConfig

    def run(self):
        return self.config.value * 2


@dataclass
class LeafC(AbstractModule):
    config: LeafCConfig

    def run(self):
        return self.config.name.upper()


class ParentConfig(AbstractConfig):
    type: Literal["B"] = "B"
    child: Annotated[Union[LeafAConfig, LeafCConfig], Field(discriminator="type")]
    children: List[Annotated[Union[LeafAConfig, LeafCConfig], Field(discriminator="type")]] = []

    def build(self):
        return Parent(config=self, child=self.child.build(), children=[c.build() for c in self.children])


@dataclass
class Parent(AbstractModule):
    config: ParentConfig
    child: AbstractModule = None
    children: list = None


class HandWrittenInit(AbstractModule):
    def __init__(self, config):
        self.config = config


class HandWrittenConfig(AbstractConfig):
    type: Literal["H"] = "H"

    def build(self):
        return HandWrittenInit(self)


@pytest.fixture
def reg():
    r = Register()
    r.registrations.clear()
    return r


def _write(tmp_path, name, payload):
    p = tmp_path / name
    p.write_text(payload if isinstance(payload, str) else json.dumps(payload))
    return str(p)


# ---------------------------------------------------------------- config tag enforcement (config_test.py)
def test_discriminated_union_parses_by_tag():
    p = ParentConfig(**{"type": "B", "child": {"type": "C", "name": "x"}})
    assert isinstance(p.child, LeafCConfig) and p.child.name == "x"
    p = ParentConfig(**{"type": "B", "child": {"type": "A", "value": 7}})
    assert isinstance(p.child, LeafAConfig) and p.child.value == 7


def test_wrong_tag_is_a_validation_error():
    with pytest.raises(ValidationError) as e:
        LeafAConfig(**{"type": "Z"})
    assert "Input should be 'A'" in str(e.value)


def test_missing_type_declaration_is_a_type_error():
    with pytest.raises(TypeError):
        class NoTag(AbstractConfig):  # noqa: F841
            value: int = 0


def test_type_must_be_single_literal_with_matching_default():
    with pytest.raises(TypeError):
        class NotLiteral(AbstractConfig):  # noqa: F841
            type: str = "x"
    with pytest.raises(TypeError):
        class WrongDefault(AbstractConfig):  # noqa: F841
            type: Literal["X"] = "Y"
    with pytest.raises(TypeError):
        class Inherited(LeafAConfig):  # noqa: F841  (must re-declare the tag in its own body)
            extra: int = 0


# ---------------------------------------------------------------- config -> module chains (module_test.py)
def test_build_returns_module_holding_its_config():
    m = LeafAConfig(value=21).build()
    assert isinstance(m, AbstractModule) and m.config.value == 21 and m.run() == 42


def test_nested_build():
    m = ParentConfig(**{"type": "B", "child": {"type": "A", "value": 3}}).build()
    assert isinstance(m.child, LeafA) and m.child.run() == 6


def test_nested_build_other_branch():
    m = ParentConfig(**{"type": "B", "child": {"type": "C", "name": "ab"}}).build()
    assert isinstance(m.child, LeafC) and m.child.run() == "AB"


def test_list_of_discriminated_children():
    m = ParentConfig(**{"type": "B", "child": {"type": "A"},
                        "children": [{"type": "A", "value": 2}, {"type": "C", "name": "q"}, {"type": "A"}]}).build()
    assert [type(c).__name__ for c in m.children] == ["LeafA", "LeafC", "LeafA"]
    assert [c.run() for c in m.children] == [4, "Q", 2]


def test_abstract_build_raises():
    with pytest.raises(NotImplementedError):
        AbstractConfig.build(LeafAConfig())


# ---------------------------------------------------------------- registry (register_test.py)
def test_register_is_a_singleton(reg):
    assert Register() is reg


def test_register_valid_config(reg, tmp_path):
    reg.register(_write(tmp_path, "a.json", {"type": "A", "value": 5}), "app", LeafAConfig)
    assert reg.get_object("app").run() == 10


def test_register_hand_written_module(reg, tmp_path):
    reg.register(_write(tmp_path, "h.json", {"type": "H"}), "h", HandWrittenConfig)
    assert isinstance(reg.get_object("h"), HandWrittenInit)


def test_register_invalid_config_prints_and_skips(reg, tmp_path):
    with patch("builtins.print") as pr:
        reg.register(_write(tmp_path, "bad.json", {"type": "A", "value": "not-an-int"}), "bad", LeafAConfig)
    assert pr.called and "Error registering bad" in pr.call_args[0][0]
    assert "bad" not in reg.registrations


def test_register_wrong_tag_prints_and_skips(reg, tmp_path):
    with patch("builtins.print") as pr:
        reg.register(_write(tmp_path, "t.json", {"type": "C"}), "t", LeafAConfig)
    assert pr.called and "t" not in reg.registrations


def test_register_missing_file_raises(reg):
    with pytest.raises(FileNotFoundError):
        reg.register("/nonexistent/definitely.json", "x", LeafAConfig)


def test_register_empty_file_prints(reg, tmp_path):
    with patch("builtins.print") as pr:
        reg.register(_write(tmp_path, "e.json", ""), "e", LeafAConfig)
    assert pr.called and "e" not in reg.registrations


def test_register_malformed_json_prints(reg, tmp_path):
    with patch("builtins.print") as pr:
        reg.register(_write(tmp_path, "m.json", "{not json"), "m", LeafAConfig)
    assert pr.called and "m" not in reg.registrations


def test_register_overwrites_same_name(reg, tmp_path):
    reg.register(_write(tmp_path, "1.json", {"type": "A", "value": 1}), "app", LeafAConfig)
    first = reg.get_object("app")
    reg.register(_write(tmp_path, "2.json", {"type": "A", "value": 9}), "app", LeafAConfig)
    assert reg.get_object("app") is not first and reg.get_object("app").run() == 18


def test_register_multiple_types(reg, tmp_path):
    reg.register(_write(tmp_path, "a.json", {"type": "A"}), "a", LeafAConfig)
    reg.register(_write(tmp_path, "c.json", {"type": "C", "name": "z"}), "c", LeafCConfig)
    reg.register(_write(tmp_path, "b.json", {"type": "B", "child": {"type": "C"}}), "b", ParentConfig)
    assert reg.get_object("a").run() == 2 and reg.get_object("c").run() == "Z"
    assert isinstance(reg.get_object("b").child, LeafC)


def test_get_object_unknown_name_is_key_error(reg):
    with pytest.raises(KeyError):
        reg.get_object("nope")


def test_singleton_decorator_ignores_later_arguments():
    @singleton
    class Box:
        def __init__(self, v=0):
            self.v = v

    assert Box(3) is Box(4) and Box().v == 3
