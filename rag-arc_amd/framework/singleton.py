"""Class decorator: one instance per decorated class (reference: framework/singleton_decorator.py:1-6)."""


def singleton(cls):
    made = {}

    def get(*args, **kwargs):
        if cls not in made:
            made[cls] = cls(*args, **kwargs)
        return made[cls]

    get.__wrapped__ = cls
    return get
