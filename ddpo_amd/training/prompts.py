"""`prompt_fn` plugin surface: prompt generators looked up by name.

Behavioural mirror of /root/reference/ddpo/training/prompts.py: `make_prompts(fn_name, batch_size, identical_batch,
**kwargs)` (:29-34) resolves `fn_name` in this module's namespace and returns (inference_prompts: list[str],
training_prompts: tuple[list[str]], metadata: tuple[dict]).  Every generator draws from Python's global `random`
state in the same ORDER and with the same calls as the reference (randint / choice), because that order is what
makes the prompt stream reproducible from the seed.  User plugins: define a function here (or
`register` one) with the signature  fn(**prompt_kwargs, evaluate=<bool>) -> (str, list[str], dict).
"""
import functools
import os
import random

_REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _resolve(path):
    return path if os.path.isabs(path) or os.path.exists(path) else os.path.join(_REPO, path)


@functools.lru_cache(maxsize=None)
def load_lines(loadpath):
    """One prompt per line, stripped (reference ddpo/utils/serialization.py:510-518)."""
    with open(_resolve(loadpath), "r") as f:
        return [line.strip() for line in f.readlines()]


@functools.lru_cache(maxsize=None)
def load_general_prompts(loadpath):
    """VQA-style prompt files: blocks of 'PROMPT: ...' followed by 'SUB|VERB|OBJ Q:' / 'A:' pairs
    (reference ddpo/utils/serialization.py:483-507)."""
    entries, cur = [], None
    with open(_resolve(loadpath), "r") as f:
        for raw in f:
            line = raw.strip()
            if line.startswith("PROMPT: "):
                cur = {"prompt": line[len("PROMPT: "):].strip(), "questions": [], "answers": []}
                entries.append(cur)
            elif cur is not None and " Q: " in line[:8]:
                cur["questions"].append(line.split(" Q: ", 1)[1].strip())
            elif cur is not None and " A: " in line[:8]:
                cur["answers"].append(line.split(" A: ", 1)[1].strip())
    return entries


class _ImageNet:
    """ImageNet-1k label strings (assets/imagenet_labels.txt, index = line) and the colour list."""

    @property
    @functools.lru_cache(maxsize=None)
    def classes(self):
        return load_lines("assets/imagenet_labels.txt")

    @property
    @functools.lru_cache(maxsize=None)
    def colors(self):
        return load_lines("assets/color_names.txt")


imagenet = _ImageNet()


# minimal stand-ins for the two `inflect` calls the reference makes (inflect is not installable here)
def _a(noun):
    return ("an " if noun[:1].lower() in "aeiou" else "a ") + noun


_NUMBER_WORDS = ["zero", "one", "two", "three", "four", "five", "six", "seven", "eight", "nine", "ten", "eleven", "twelve"]


def _number_to_words(n):
    return _NUMBER_WORDS[n] if 0 <= n < len(_NUMBER_WORDS) else str(n)


def _plural(noun):
    irregular = {"mouse": "mice", "goose": "geese", "sheep": "sheep", "deer": "deer", "fish": "fish", "wolf": "wolves", "fox": "foxes"}
    if noun in irregular:
        return irregular[noun]
    if noun.endswith(("s", "x", "z", "ch", "sh")):
        return noun + "es"
    if noun.endswith("y") and noun[-2:-1] not in "aeiou":
        return noun[:-1] + "ies"
    return noun + "s"


# --------------------------------- general api --------------------------------- #
def register(fn):
    """Expose a user-defined prompt function under its own name."""
    globals()[fn.__name__] = fn
    return fn


def make_prompts(fn_name, batch_size, identical_batch=False, **kwargs):
    prompt_fn = globals()[fn_name]
    if identical_batch:
        prompt, training, meta = prompt_fn(**kwargs)
        return [prompt] * batch_size, [training] * batch_size, [meta] * batch_size
    drawn = [prompt_fn(**kwargs) for _ in range(batch_size)]
    prompts, training, meta = zip(*drawn)
    return list(prompts), training, meta


# ---------------------------- specific experiments ---------------------------- #
def _single(prompt, meta=None):
    return prompt, [prompt], ({} if meta is None else meta)


def _pick(training_prompts, meta=None):
    """The reference draws the inference prompt with random.choice even from a 1-element list (consumes RNG state)."""
    return random.choice(training_prompts), training_prompts, ({} if meta is None else meta)


def get_random_class(idx=None, low=None, high=None):
    if idx is not None:
        return imagenet.classes[idx]
    if low is not None and high is not None:
        return imagenet.classes[random.randint(low, high)]
    return random.choice(imagenet.classes)


def person_pet(evaluate=False):
    return _pick(["a photo of a person with their pet"])


def consistent_animals(evaluate=False):
    return _single("a husky and a shoebill stork on the beach in a single image")


def consistent_imagenet_animals(colors=False):
    c1, c2 = get_random_class(), get_random_class()
    if colors:
        return _single(f"a realistic photo of a {random.choice(imagenet.colors)} {c1} and a {random.choice(imagenet.colors)} {c2}")
    return _single(f"a realistic photo of a {c1} and a {c2}")


def consistent_imagenet_animals_3(colors=False):
    c1, c2, c3 = get_random_class(), get_random_class(), get_random_class()
    if colors:
        k1, k2, k3 = (random.choice(imagenet.colors) for _ in range(3))
        return _single(f"a realistic photo of a {k1} {c1}, a {k2} {c2}, and a {k3} {c3}")
    return _single(f"a realistic photo of a {c1}, a {c2}, and a {c3}")


def n_fingers(evaluate=False):
    n = random.randint(1, 4)
    return _single(f'a photo of a hand holding up {n} finger{"s" if n > 1 else ""}')


def imagenet_single(evaluate=False, idx=None):
    return _single(f"a realistic photo of a {get_random_class(idx=idx)}")


def imagenet_aesthetic(evaluate=False):
    return _pick([f"a realistic photo of a {get_random_class()}"])


def imagenet_simple(evaluate=False, idx=None):
    return _single(f"a {get_random_class(idx=idx)}")


def imagenet_dogs(evaluate=False, idx=None):
    return _pick([f"{get_random_class(idx=idx, low=151, high=268)}"])


simple_dogs = imagenet_dogs


def animal_debug(evaluate=False, idx=None):
    return _pick(["a peacock"])


def imagenet_animals(evaluate=False, idx=None):
    return _pick([f"{get_random_class(idx=idx, low=0, high=397)}"])


def from_file(loadpath, evaluate=False, idx=None):
    prompts = load_lines(loadpath)
    return _single(prompts[idx] if idx is not None else random.choice(prompts))


def vqa_dataset(loadpath, max_samples=None, evaluate=False):
    entry = random.choice(load_general_prompts(loadpath))
    return entry["prompt"], [entry["prompt"]], entry


def manual(prompts, evaluate=False):
    return _pick(prompts)


def nouns_activities(nouns_path, activities_path, evaluate=False):
    nouns, activities = load_lines(nouns_path), load_lines(activities_path)
    noun = random.choice(nouns)                 # noun first, then activity: evaluation order of the reference f-string
    activity = random.choice(activities)
    return _single(f"{_a(noun)} {activity}")


def counting(nouns_path, number_range, evaluate=False):
    nouns = load_lines(nouns_path)
    number = _number_to_words(random.randint(*number_range))
    noun = random.choice(nouns)
    plural_noun = _plural(noun)
    meta = {"questions": [f"How many {plural_noun} are there in this image?", "What animal is in this image?"],
            "answers": [number, noun]}
    return _single(f"{number} {plural_noun}", meta)
